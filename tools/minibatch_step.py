#!/usr/bin/env python3
"""The folded train step on a small gathered minibatch (the reference's default --batch-size 16), replayed from a hipGraph:
   python tools/minibatch_step.py [--batch 16] [--items 1000] [--ability-dim 8] [--reps 500]     (for tools/prof_cmd.sh / timing)"""
import argparse, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd.torch_core.models import VIBO_2PL
from vibo_amd.trainer import FusedTrainer
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--items', type=int, default=1000)
ap.add_argument('--ability-dim', type=int, default=8)
ap.add_argument('--persons', type=int, default=20000)
ap.add_argument('--reps', type=int, default=500)
a = ap.parse_args()
d = torch.device('cuda:0')
g = torch.Generator(device=d).manual_seed(0)
resp = (torch.rand(a.persons, a.items, device=d, generator=g) < 0.5).float()
mask = torch.rand(a.persons, a.items, device=d, generator=g) >= 0.1
torch.manual_seed(1)
model = VIBO_2PL(a.ability_dim, a.items, ability_merge='product').to(d)
tr = FusedTrainer(model, lr=5e-3, rng='native', seed=3)
rows = torch.randperm(a.persons, device=d)[:a.batch].contiguous()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        tr.step(resp, mask, row_index=rows)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    tr.step(resp, mask, row_index=rows)
for _ in range(20):
    gr.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    gr.replay()
torch.cuda.synchronize()
print(f'B={a.batch} I={a.items} A={a.ability_dim}: {(time.perf_counter() - t0) / a.reps * 1e6:.1f} us per replayed step')
