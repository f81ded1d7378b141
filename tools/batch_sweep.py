#!/usr/bin/env python3
"""Train-step throughput vs minibatch size on a resident matrix (SURVEY.md §8d asks for B = 16, 4096, 65536 and the
full shard).  One FusedTrainer step per minibatch (row-index gather in the kernel), replayed from a hipGraph.
   python tools/batch_sweep.py [--persons P] [--items I] [--ability-dim A]"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd.torch_core.models import VIBO_2PL
from vibo_amd.trainer import FusedTrainer

ap = argparse.ArgumentParser()
ap.add_argument('--persons', type=int, default=1_000_000)
ap.add_argument('--items', type=int, default=1000)
ap.add_argument('--ability-dim', type=int, default=1)
ap.add_argument('--batches', type=int, nargs='*', default=[16, 256, 4096, 65536, 0])
a = ap.parse_args()
d = torch.device('cuda:0')
g = torch.Generator(device=d).manual_seed(0)
P, I, A = a.persons, a.items, a.ability_dim
resp = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
mask = torch.rand(P, I, device=d, generator=g) >= 0.1
for B in a.batches:
    B = B or P
    torch.manual_seed(0)
    model = VIBO_2PL(A, I, ability_merge='product').to(d)
    tr = FusedTrainer(model, lr=5e-3, rng='native', seed=1)
    rows = torch.randperm(P, device=d)[:B].contiguous() if B < P else None
    for mode in ('eager', 'graph'):
        step = lambda: tr.step(resp, mask, row_index=rows)
        if mode == 'graph':
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                step()
            step = gr.replay
        n = max(20, min(2000, int(2e9 // (B * I)) or 20))
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f'B={B:8d} {mode:6s}: {dt * 1e6:9.1f} us/step  {B * I / dt / 1e9:9.3f} G terms/s')
