#!/usr/bin/env python3
"""Train-step time of the module path (elbo_step + backward + Adam) eager vs replayed from a hipGraph
(vibo_amd.torch_core.vibo.GraphedModuleStep), for the configurations FusedTrainer does not cover.
   python tools/step_time.py [--irt 3] [--items 1000] [--ability-dim 1] [--batch 16] [--cond] [--flows 4] [--merge product]"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd.torch_core import models as M
from vibo_amd.torch_core.vibo import GraphedModuleStep

ap = argparse.ArgumentParser()
ap.add_argument('--persons', type=int, default=20000)
ap.add_argument('--items', type=int, default=1000)
ap.add_argument('--ability-dim', type=int, default=1)
ap.add_argument('--irt', type=int, default=3)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--cond', action='store_true')
ap.add_argument('--flows', type=int, default=0)
ap.add_argument('--merge', default='product')
a = ap.parse_args()
d = torch.device('cuda:0')
g = torch.Generator(device=d).manual_seed(0)
P, I, A, B = a.persons, a.items, a.ability_dim, a.batch


class Data:
    pass


data = Data()
data.response = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
data.mask = torch.rand(P, I, device=d, generator=g) >= 0.1
data.device = d
cls = {1: M.VIBO_1PL, 2: M.VIBO_2PL, 3: M.VIBO_3PL}[a.irt]
for mode in ('eager', 'graph'):
    torch.manual_seed(0)
    model = cls(A, I, ability_merge=a.merge, conditional_posterior=a.cond, n_norm_flows=a.flows).to(d)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, capturable=mode == 'graph', fused=True)      # (as the CLI builds it)
    step = GraphedModuleStep(model, opt, data, B)
    if mode == 'eager':
        step.WARMUP = 1 << 30
    rows = [torch.randperm(P, device=d)[:B].contiguous() for _ in range(8)]
    for k in range(6):
        step(rows[k % 8], 1.0)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for k in range(n):
        loss = step(rows[k % 8], 1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'irt={a.irt} I={I} A={A} B={B} cond={a.cond} flows={a.flows} merge={a.merge} {mode:6s}: {dt * 1e6:9.1f} us/step  loss {float(loss):.1f}')

# the same configuration through the fused trainer (FusedTrainer / FusedCondFlowTrainer), eager and replayed
from vibo_amd.trainer import FusedTrainer, fused_trainer_covers
torch.manual_seed(0)
model = cls(A, I, ability_merge=a.merge, conditional_posterior=a.cond, n_norm_flows=a.flows).to(d)
if fused_trainer_covers(model):
    tr = FusedTrainer(model, lr=5e-3, rng='native', seed=3)
    rbuf = torch.zeros(B, dtype=torch.int64, device=d)
    rows = [torch.randperm(P, device=d)[:B].contiguous() for _ in range(8)]
    for k in range(4):
        rbuf.copy_(rows[k]); tr.step(data.response, data.mask, row_index=rbuf)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for k in range(n):
        rbuf.copy_(rows[k % 8]); loss = tr.step(data.response, data.mask, row_index=rbuf)
    torch.cuda.synchronize()
    print(f'fused trainer ({type(tr).__name__}) eager : {(time.perf_counter() - t0) / n * 1e6:9.1f} us/step  loss {float(loss):.1f}')
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        loss = tr.step(data.response, data.mask, row_index=rbuf)
    for k in range(5):
        gr.replay()
    torch.cuda.synchronize()
    n = 1000
    t0 = time.perf_counter()
    for k in range(n):
        rbuf.copy_(rows[k % 8]); gr.replay()
    torch.cuda.synchronize()
    print(f'fused trainer ({type(tr).__name__}) graph : {(time.perf_counter() - t0) / n * 1e6:9.1f} us/step  loss {float(loss):.1f}')
