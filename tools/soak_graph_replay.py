#!/usr/bin/env python3
"""Soak test of the default training path: the fused train step replayed from a hipGraph for thousands of steps must stay
BITWISE equal to the same steps launched eagerly (native Philox noise is a pure function of the step counter).
   python tools/soak_graph_replay.py [--steps 3000] [--batch 16] [--items 100] [--ability-dim 1]"""
import argparse
import copy
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd.torch_core.models import VIBO_1PL, VIBO_2PL, VIBO_3PL
from vibo_amd.trainer import FusedTrainer

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=3000)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--items', type=int, default=100)
ap.add_argument('--ability-dim', type=int, default=1)
ap.add_argument('--persons', type=int, default=8000)
ap.add_argument('--irt', type=int, default=2)
ap.add_argument('--cond', action='store_true', help='--conditional-posterior (FusedCondFlowTrainer)')
ap.add_argument('--flows', type=int, default=0, help='--n-norm-flows')
a = ap.parse_args()
d = torch.device('cuda:0')
g = torch.Generator(device=d).manual_seed(0)
P, I, A, B = a.persons, a.items, a.ability_dim, a.batch
resp = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
mask = torch.rand(P, I, device=d, generator=g) >= 0.1
torch.manual_seed(1)
m1 = {1: VIBO_1PL, 2: VIBO_2PL, 3: VIBO_3PL}[a.irt](A, I, ability_merge='product', conditional_posterior=a.cond, n_norm_flows=a.flows).to(d)
m2 = copy.deepcopy(m1)
t1 = FusedTrainer(m1, lr=5e-3, rng='native', seed=7)
t2 = FusedTrainer(m2, lr=5e-3, rng='native', seed=7, fold=False)      # the four-launch form, launched eagerly: the yardstick
rows = torch.zeros(B, dtype=torch.int64, device=d)
perm = torch.randperm(P, device=d)
for _ in range(3):
    rows.copy_(perm[:B]); t1.step(resp, mask, row_index=rows); t2.step(resp, mask, row_index=rows)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    lg = t1.step(resp, mask, row_index=rows)
bad = 0
for it in range(a.steps):
    s = (it * B) % (P - B)
    rows.copy_(perm[s:s + B])
    gr.replay()
    l2 = t2.step(resp, mask, row_index=rows)
    if it % 250 == 0 or it == a.steps - 1:
        same = torch.equal(lg, l2) and all(torch.equal(x, y) for x, y in zip(m1.state_dict().values(), m2.state_dict().values()))
        print(f'step {it}: loss {float(lg):.4f} {"bitwise equal" if same else "MISMATCH"}', flush=True)
        bad += 0 if same else 1
print('soak', 'ok' if bad == 0 else f'FAILED ({bad} mismatches)')
sys.exit(1 if bad else 0)
