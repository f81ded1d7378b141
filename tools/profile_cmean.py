#!/usr/bin/env python3
"""Time vibo_code_table_sum_forward / _backward (--ability-merge mean + --conditional-posterior: one-hot [B, 2I] x [2I, 64] on
the matrix pipe) against the round-2 formulation (two fp32 GEMMs on materialised indicator matrices).
   python tools/profile_cmean.py [--persons 200000] [--items 1000]"""
import argparse, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument('--persons', type=int, default=200000)
ap.add_argument('--items', type=int, default=1000)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--no-gemm', action='store_true')
a = ap.parse_args()
d = torch.device('cuda:0')
g = torch.Generator(device=d).manual_seed(0)
B, I = a.persons, a.items
resp = (torch.rand(B, I, device=d, generator=g) < 0.5).float()
mask = torch.rand(B, I, device=d, generator=g) >= 0.1
cc = ops.pack_cell_codes(resp, mask)
feat = torch.randn(2, I, 64, device=d, generator=g).requires_grad_(True)
gout = torch.randn(B, 64, device=d, generator=g)


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


def native():
    S = ops.CodeTableSumFn.apply(feat, cc)
    torch.autograd.grad(S, feat, gout)


def gemm():
    obs = mask.float()
    right = resp * obs
    S = obs @ feat[0] + right @ (feat[1] - feat[0])
    torch.autograd.grad(S, feat, gout)


flop = 2 * 2.0 * B * 2 * I * 64          # forward + backward, dense-equivalent
t = timed(native)
print(f'B={B} I={I}: native fwd+bwd {t:.3f} ms = {B * I / t / 1e6:.1f} G cells/s, {flop / t / 1e9:.1f} dense-equivalent TFLOP/s (x2 f16 passes issued)')
if not a.no_gemm:
    t2 = timed(gemm)
    print(f'          two fp32 GEMMs on indicator matrices (round 2) {t2:.3f} ms  -> {t2 / t:.1f} x')
