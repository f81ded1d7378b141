#!/usr/bin/env python3
"""Train-step and in-situ kernel time of the folded step (one hipGraph per step) over a list of shapes, for same-box A/B runs of
library builds: `VIBO_HIP_LIB=<variant .so> python tools/ab_insitu.py 100000x1000x1 125000x1000x8 ...` (persons x items x
ability_dim[xc for cell codes]).  Kernel time = the matrix kernel's own stamps (ops.InsituTimer, vibo_set_insitu_timer), step
time = wall clock over the replays.  One line per shape: step us | kernel us (mean, min) over `--reps` replays, best of 3 rounds."""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd import _lib, ops
from vibo_amd.torch_core.models import VIBO_2PL
from vibo_amd.trainer import FusedTrainer

ap = argparse.ArgumentParser()
ap.add_argument('shapes', nargs='+')
ap.add_argument('--reps', type=int, default=200)
ap.add_argument('--tag', default=os.path.basename(os.environ.get('VIBO_HIP_LIB', 'cur')))
a = ap.parse_args()
d = torch.device('cuda:0')
have_timer = hasattr(_lib.load(), 'vibo_set_insitu_timer')
for sh in a.shapes:
    f = sh.split('x')
    P, I, A = int(f[0]), int(f[1]), int(f[2])
    codes = len(f) > 3 and f[3] == 'c'
    g = torch.Generator(device=d).manual_seed(0)
    resp = (torch.rand(P, I, device=d, generator=g) < 0.5).float()
    mask = torch.rand(P, I, device=d, generator=g) >= 0.1
    if codes:
        resp, mask = ops.pack_cell_codes(resp, mask), None
    torch.manual_seed(1)
    model = VIBO_2PL(A, I, ability_merge='product').to(d)
    tr = FusedTrainer(model, lr=5e-3, rng='native', seed=3)
    tm = ops.InsituTimer(d) if have_timer else None
    if tm:
        tm.arm()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            tr.step(resp, mask)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        loss = tr.step(resp, mask)
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    best = None
    for rnd in range(3):
        if tm:
            tm.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            gr.replay()
        torch.cuda.synchronize()
        step_us = (time.perf_counter() - t0) / a.reps * 1e6
        k = tm.read() if tm else {}
        cur = (step_us, k.get('mean_ms', 0) * 1e3, k.get('min_ms', 0) * 1e3)
        best = cur if best is None or cur[0] < best[0] else best
    if tm:
        tm.disarm()
    print(f'{a.tag:24s} {sh:20s} step {best[0]:8.1f} us   kernel {best[1]:8.1f} us (min {best[2]:.1f})   loss {float(loss):.6g}', flush=True)
    del resp, mask, model, tr, gr
    torch.cuda.empty_cache()
