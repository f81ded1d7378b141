#!/usr/bin/env python3
"""Multi-sample forward (log_marginal's loop body) vs one forward launch per sample.
   python tools/profile_multi.py [--persons P] [--items I] [--ability-dim A] [--samples S]"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

ap = argparse.ArgumentParser()
ap.add_argument('--persons', type=int, default=1_000_000)
ap.add_argument('--items', type=int, default=1000)
ap.add_argument('--ability-dim', type=int, default=1)
ap.add_argument('--samples', type=int, default=16)
ap.add_argument('--irt', type=int, default=2)
ap.add_argument('--codes', action='store_true', help='rows as 1-byte cell codes (VIBO_MASK_CODES)')
ap.add_argument('--gather', action='store_true')
a = ap.parse_args()
d = torch.device('cuda:0')
g = torch.Generator(device=d).manual_seed(0)
P, I, A, S = a.persons, a.items, a.ability_dim, a.samples
spec = ElboSpec(irt_model=a.irt, ability_dim=A)
resp = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
mask = torch.rand(P, I, device=d, generator=g) >= 0.1
table = torch.randn(2, 2 * A, device=d, generator=g) * 0.5
items = torch.randn(S, I, spec.item_dim, device=d, generator=g)
eps = torch.randn(S, P, A, device=d, generator=g)
m, code = ops.prepare_mask(mask)
if a.codes:
    resp, m, code = ops.prepare_rows(ops.pack_cell_codes(resp, mask), None)
ridx = torch.randperm(P, device=d) if a.gather else None


def multi():
    return ops._hip_multi_forward(spec, resp, m, code, ridx, table, items, eps, None, _lib.REG_SAMPLED, P)


def singles():
    return [ops._hip_launch_elbo(spec, resp, m, code, ridx, table, items[s], eps[s], None, _lib.REG_SAMPLED, False, P).scalars
            for s in range(S)]


for name, f in (('multi-sample kernel', multi), ('one launch per sample', singles)):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'P={P} I={I} A={A} S={S} codes={a.codes} gather={a.gather} {name:24s}: {dt * 1e3:8.3f} ms = {dt * 1e3 / S:6.3f} ms per sample, '
          f'{P * I * S / dt / 1e12:.3f} T sample-terms/s')
