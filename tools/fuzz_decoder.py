#!/usr/bin/env python3
"""Randomised parity sweep of the per-term MLP decoder kernel (vibo_decoder_fwd_bwd) and the planar-flow stack kernels against
float64 autograd of the same formulas.   python tools/fuzz_decoder.py [--seconds 120] [--seed 0]     (needs the MI355X)"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from vibo_amd import decoder as D
from vibo_amd import ops
from test_gpu_decoder import torch_reference, rel

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=120)
ap.add_argument('--seed', type=int, default=0)
a = ap.parse_args()
rng = random.Random(a.seed)
dev = torch.device('cuda:0')
t0, n, worst = time.time(), 0, 0.0
while time.time() - t0 < a.seconds:
    mode = rng.choice(['deep', 'residual', 'link', 'residual3', 'link3'])
    B = rng.choice([1, 2, 3, 16, 33, 64, 130, 257])
    I = rng.choice([1, 5, 16, 63, 64, 65, 100, 200, 333])
    missing = rng.choice([0.0, 0.0, 0.2, 0.6])
    seed = rng.randrange(1 << 30)
    g = torch.Generator().manual_seed(seed)
    H = 64
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).double()
    resp = (torch.rand(B, I, generator=g) < 0.5).double()
    mask = (torch.rand(B, I, generator=g) >= missing).double() if missing > 0 else None
    # (outputs stay out of the probability-clamp band |o| > 15.9: there the clamp decision flips with fp32 rounding, in the
    #  reference as much as here, and a float64 oracle cannot arbitrate -- DESIGN.md section 4)
    sc = rng.choice([0.3, 1.0])
    t = dict(V=rn(B, H, sc=0.7 * sc), W2=rn(H, H, sc=0.18), b2=rn(H, sc=0.1), w3=rn(H, sc=0.25), b3=rn(1, sc=0.1))
    t['U'] = rn(I, H, sc=0.7) if not mode.startswith('link') else None
    t['logit'] = rn(B, I, sc=1.5 * sc) if mode != 'deep' else None
    t['w1'] = rn(H, sc=0.5) if mode.startswith('link') else None
    t['guess'] = torch.sigmoid(rn(I, sc=0.5) - 1.0) if mode.endswith('3') else None      # (guess <= ~0.7: the mixture's clamp band starts at (1 - g)(1 - sigma) < eps32)
    resid = 1.0 if mode.startswith('residual') else 0.0
    leaves = {k: v.clone().requires_grad_(True) for k, v in t.items() if v is not None}
    ll_ref, _ = torch_reference(resp, mask, leaves.get('U'), leaves['V'], leaves['W2'], leaves['b2'], leaves['w3'], leaves['b3'],
                                leaves.get('logit'), leaves.get('w1'), leaves.get('guess'), resid)
    ll_ref.backward()
    dl = {k: v.float().to(dev).requires_grad_(True) for k, v in t.items() if v is not None}
    ll = D.decoder_log_lik(resp.float().to(dev), mask.bool().to(dev) if mask is not None else None, U=dl.get('U'), V=dl['V'], W2=dl['W2'],
                           b2=dl['b2'], w3=dl['w3'], b3=dl['b3'], logit=dl.get('logit'), w1=dl.get('w1'), guess=dl.get('guess'), resid=resid)
    ll.backward()
    nobs = float(mask.sum()) if mask is not None else B * I
    errs = {'ll': abs(float(ll.detach()) - float(ll_ref.detach())) / max(abs(float(ll_ref.detach())), 0.05 * max(nobs, 1.0))}
    for k in leaves:
        if float(leaves[k].grad.abs().max()) > 1e-6:
            errs[k] = rel(dl[k].grad.double().cpu(), leaves[k].grad)
            if k == 'b3':      # one scalar = a sum of nobs terms of either sign: measured against the sum's natural scale, not against
                #               a value that may cancel to ~0 (seen: 0.0027 from 2 145 terms of ~0.5, 6e-6 apart = 2.4e-3 "relative")
                errs[k] = float((dl[k].grad.double().cpu() - leaves[k].grad).abs().max()) / max(float(leaves[k].grad.abs().max()), 0.01 * max(nobs, 1.0) ** 0.5)
    # (a handful of terms with strongly negative pre-activations: ELU' = h + 1 carries the 6e-8 absolute rounding of h, exactly as
    #  the reference's in-place ELU backward does; with thousands of terms it averages out)
    gtol = 3e-3 if B * I < 64 else 5e-4
    bad = {k: v for k, v in errs.items() if not (v < (5e-5 if k == 'll' else gtol))}
    worst = max(worst, max(errs.values()))
    n += 1
    if bad:
        detail = {k: (float(dl[k].grad.double().cpu().abs().max()), float(leaves[k].grad.abs().max())) for k in bad if k != 'll'}
        print(f'FAIL mode={mode} B={B} I={I} missing={missing} scale={sc} seed={seed}: {bad}  (|grad| max: kernel, float64 reference) {detail}')
        sys.exit(1)
    # the flow stack on a random small matrix
    N, Dm, K = rng.choice([1, 7, 256, 257, 1000]), rng.choice([1, 2, 3, 9, 10]), rng.choice([1, 2, 4, 8])
    u, w = rn(K, Dm, sc=0.5), rn(K, Dm, sc=0.5)      # (N(0,1) parameters in 10 dims make 8 stacked flows expansive: rounding errors grow ~10x per flow)
    uw = (u * w).sum(1, keepdim=True)
    uhat = u + (torch.nn.functional.softplus(uw) - 1.0 - uw) * w / (w * w).sum(1, keepdim=True)
    z = rn(N, Dm).requires_grad_(True)
    packed = torch.cat([uhat, w, rn(K, 1)], 1).requires_grad_(True)
    zz, total = z, 0.0
    for k in range(K):
        th = torch.tanh(zz @ packed[k, Dm:2 * Dm] + packed[k, 2 * Dm])
        zz = zz + packed[k, :Dm].unsqueeze(0) * th.unsqueeze(1)
        total = total + torch.log(torch.abs(1.0 + (1.0 - th * th) * torch.dot(packed[k, Dm:2 * Dm], packed[k, :Dm])) + 1e-8)
    cz, cl = rn(N, Dm), rn(N)
    ((zz * cz).sum() + (total * cl).sum()).backward()
    z32, p32 = z.detach().float().to(dev).requires_grad_(True), packed.detach().float().to(dev).requires_grad_(True)
    zo, la = ops.FlowStackFn.apply(z32, p32)
    ((zo * cz.float().to(dev)).sum() + (la * cl.float().to(dev)).sum()).backward()
    ferr = {'z': float((zo.detach().double().cpu() - zz.detach()).abs().max()) / max(1.0, float(zz.abs().max())),
            'ladj': float((la.detach().double().cpu() - total.detach()).abs().max()) / max(1.0, float(total.abs().max())),
            'g_z': rel(z32.grad.double().cpu(), z.grad), 'g_p': rel(p32.grad.double().cpu(), packed.grad)}
    # (d ladj = 1 / psi with psi = 1 + (1 - t^2) w.uhat >= 0 only just: rows near psi = 0 carry fp32 noise into the parameter sums)
    fbad = {k: v for k, v in ferr.items() if not (v < (2e-5 if k == 'z' else 2e-4 if k == 'ladj' else 2e-3))}
    worst = max(worst, max(ferr.values()))
    if fbad:
        print(f'FAIL flow N={N} D={Dm} K={K} seed={seed}: {fbad}')
        sys.exit(1)
print(f'fuzz decoder ok: {n} decoder + {n} flow-stack configurations, worst relative error {worst:.2e}')
