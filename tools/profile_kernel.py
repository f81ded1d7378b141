#!/usr/bin/env python3
"""Launch the fused ELBO kernel a few times on a synthetic matrix (for rocprofv3).
   python tools/profile_kernel.py [--persons P] [--items I] [--ability-dim A] [--iters N]"""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

ap = argparse.ArgumentParser()
ap.add_argument('--persons', type=int, default=1_000_000)
ap.add_argument('--items', type=int, default=1000)
ap.add_argument('--ability-dim', type=int, default=8)
ap.add_argument('--irt', type=int, default=2)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--no-grad', action='store_true')
ap.add_argument('--cond', action='store_true')
ap.add_argument('--flows', type=int, default=0)
ap.add_argument('--missing', type=float, default=0.1)
ap.add_argument('--gather', action='store_true', help='rows through a random row_index permutation (shuffled minibatch)')
ap.add_argument('--given', action='store_true', help='caller-supplied per-person posterior (VIBO_POSTERIOR_GIVEN, the --ability-merge mean path)')
ap.add_argument('--codes', action='store_true', help='rows as 1-byte cell codes (VIBO_MASK_CODES, Format P) instead of fp32 + mask')
ap.add_argument('--kernel', choices=['auto', 'matrix', 'valu'], default='auto', help='pin a row-split kernel (vibo_desc.flags)')
ap.add_argument('--cond-valu', action='store_true', help='conditional posterior: the VALU passes (VIBO_FLAG_COND_VALU) instead of the matrix-pipe ones')
ap.add_argument('--cond-three-pass', action='store_true', help='conditional posterior at ability_dim 1: the separate first pass (VIBO_FLAG_COND_THREE_PASS) instead of the one folded into the matrix kernel')
ap.add_argument('--item-scale', type=float, default=1.0, help='scale of the N(0,1) item sample (larger: more logits past the Bernoulli clamp, i.e. more tiles on the slow path)')
ap.add_argument('--init-like', action='store_true', help='item sample as a freshly initialised model draws it: mu + exp(.5 logvar) eps with mu, logvar, eps ~ N(0,1)')
ap.add_argument('--cached-rows', type=int, default=0, help='gather rows from the first N rows only (L2-resident): compute-only timing')
a = ap.parse_args()
ops.DESC_FLAGS = {'auto': 0, 'matrix': _lib.FLAG_KERNEL_MATRIX, 'valu': _lib.FLAG_KERNEL_VALU}[a.kernel] | (_lib.FLAG_COND_VALU if a.cond_valu else 0) | (_lib.FLAG_COND_THREE_PASS if a.cond_three_pass else 0)
d = torch.device('cuda:0')
g = torch.Generator(device=d).manual_seed(0)
P, I, A = a.persons, a.items, a.ability_dim
D = {1: 1, 2: A + 1, 3: A + 2}[a.irt]
resp = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
mask = torch.rand(P, I, device=d, generator=g) >= a.missing
spec = ElboSpec(irt_model=a.irt, ability_dim=A, conditional=a.cond, n_flows=a.flows, given=a.given)
table = torch.randn(*spec.table_shape(I, P), device=d, generator=g) * 0.5
flow = torch.randn(a.flows, 2 * A + 1, device=d, generator=g) * 0.5 if a.flows else None
reg = _lib.REG_SAMPLED if a.flows else _lib.REG_KL
item = torch.randn(I, D, device=d, generator=g) * a.item_scale
if a.init_like:
    item = torch.randn(I, D, device=d, generator=g) + torch.exp(0.5 * torch.randn(I, D, device=d, generator=g)) * torch.randn(I, D, device=d, generator=g)
eps = torch.randn(P, A, device=d, generator=g)
m, code = ops.prepare_mask(mask)
if a.codes:
    resp, m, code = ops.prepare_rows(ops.pack_cell_codes(resp, mask), None)
    del mask
ridx = (torch.arange(P, device=d) % a.cached_rows) if a.cached_rows else None
if a.gather:
    ridx = torch.randperm(P, device=d)
for _ in range(a.iters):
    raw = ops._hip_launch_elbo(spec, resp, m, code, ridx, table, item, eps, flow, reg, not a.no_grad, P)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    raw = ops._hip_launch_elbo(spec, resp, m, code, ridx, table, item, eps, flow, reg, not a.no_grad, P)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
print(f'P={P} I={I} A={A} irt={a.irt} given={a.given} cond={a.cond} flows={a.flows} grad={not a.no_grad} gather={ridx is not None}: {ms:.3f} ms/call, {P*I/ms/1e9:.3f} T terms/s, '
      f'{((1 if a.codes else 5)+12*A/I)*P*I/ms/1e6:.0f} GB/s algorithmic{" (1 B/cell codes)" if a.codes else ""}, ll={float(raw.scalars[0]):.1f}')
