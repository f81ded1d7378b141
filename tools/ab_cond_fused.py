#!/usr/bin/env python3
"""Conditional posterior at ability_dim 1: the matrix kernel with the first pass folded in (its XM == 3, the default where it applies)
against the three passes (VIBO_FLAG_COND_THREE_PASS) on the same inputs -- outputs side by side and the time of both.
   python tools/ab_cond_fused.py [PxI[:irt[:g]] ...]       (g = rows through a random row_index)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

d = torch.device('cuda:0')


def run(P, I, irt, gather, flags, iters, missing=0.1, grad=True, missing_mode=None):
    ops.DESC_FLAGS = flags
    g = torch.Generator(device=d).manual_seed(1)
    A = 1
    D = {1: 1, 2: A + 1, 3: A + 2}[irt]
    resp = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
    mask = torch.rand(P, I, device=d, generator=g) >= missing
    resp, mask = ops.pad_rows(resp, mask)
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=True)
    table = torch.randn(*spec.table_shape(I, P), device=d, generator=g) * 0.5
    item = torch.randn(I, D, device=d, generator=g)
    eps = torch.randn(P, A, device=d, generator=g)
    m, code = ops.prepare_mask(mask)
    ridx = torch.randperm(P, device=d, generator=g) if gather else None
    for _ in range(2):
        raw = ops._hip_launch_elbo(spec, resp, m, code, ridx, table, item, eps, None, _lib.REG_KL, grad, P)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        raw = ops._hip_launch_elbo(spec, resp, m, code, ridx, table, item, eps, None, _lib.REG_KL, grad, P)
    e1.record()
    torch.cuda.synchronize()
    return raw, e0.elapsed_time(e1) / iters


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


shapes = sys.argv[1:] or ['1000000x1000', '100000x1000', '65536x640', '4096x500:2', '33000x1000:3', '50001x998:1', '100000x1000:2:g', '40000x129']
for sh in shapes:
    parts = sh.split(':')
    P, I = (int(v) for v in parts[0].split('x'))
    irt = int(parts[1]) if len(parts) > 1 and parts[1] else 2
    gather = len(parts) > 2 and parts[2] == 'g'
    iters = max(10, min(50, int(5e9 / (P * I))))
    base = _lib.FLAG_KERNEL_MATRIX | _lib.FLAG_COND_MATRIX
    for grad in (True, False):
        f, tf = run(P, I, irt, gather, base, iters, grad=grad)
        t, tt = run(P, I, irt, gather, base | _lib.FLAG_COND_THREE_PASS, iters, grad=grad)
        errs = {'scalars': rel(f.scalars, t.scalars), 'mu': rel(f.ability_mu, t.ability_mu), 'logvar': rel(f.ability_logvar, t.ability_logvar)}
        if grad:
            n = _lib.NUM_SCALARS
            errs['d_table'] = rel(f.flat[n:n + 2 * f.n_table], t.flat[n:n + 2 * t.n_table])
            errs['d_item'] = rel(f.flat[n + 2 * f.n_table:], t.flat[n + 2 * t.n_table:])
        print(f'{sh:24s} grad={int(grad)}  fused {tf:8.3f} ms   three-pass {tt:8.3f} ms   ' + '  '.join(f'{k} {v:.1e}' for k, v in errs.items()), flush=True)
