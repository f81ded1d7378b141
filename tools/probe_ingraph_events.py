#!/usr/bin/env python3
"""Does a HIP event recorded INSIDE a captured step (torch.cuda.Event(external=True) -> event-record graph nodes) time the
fused ELBO call of the replayed step?  Compares, on the bench workload: (a) in-graph events around the native call, read
after groups of back-to-back replays; (b) bare back-to-back eager launches of the same call; (c) step period."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')]
import torch
from vibo_amd import ops
from vibo_amd.torch_core.models import VIBO_2PL
from vibo_amd.trainer import FusedTrainer

ap = argparse.ArgumentParser()
ap.add_argument('--persons', type=int, default=1_000_000)
ap.add_argument('--items', type=int, default=1000)
ap.add_argument('--ability-dim', type=int, default=8)
ap.add_argument('--batch', type=int, default=0, help='> 0: gathered minibatch of this many rows')
a = ap.parse_args()
dev = torch.device('cuda:0')
P, I, A = a.persons, a.items, a.ability_dim
g = torch.Generator(device=dev).manual_seed(1)
resp = (torch.rand(P, I, device=dev, generator=g) < 0.5).float()
mask = torch.rand(P, I, device=dev, generator=g) >= 0.1
resp = torch.where(mask, resp, torch.full_like(resp, -1.0))
torch.manual_seed(0)
model = VIBO_2PL(A, I, ability_merge='product').to(dev)
tr = FusedTrainer(model, lr=5e-3, rng='native', seed=0)
rows = torch.randperm(P, device=dev)[:a.batch].contiguous() if a.batch else None

native = ops._BACKEND['elbo']
ev = {}
last = {}
def hooked(*x, **k):
    last['a'], last['k'] = x, k
    if ev.get('on'):
        ev['e0'].record()
        out = native(*x, **k)
        ev['e1'].record()
        return out
    return native(*x, **k)
ops._BACKEND['elbo'] = hooked
step = lambda: tr.step(resp, mask, row_index=rows)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
res = {}
try:
    ev['e0'] = torch.cuda.Event(enable_timing=True, external=True)
    ev['e1'] = torch.cuda.Event(enable_timing=True, external=True)
    ev['s0'] = torch.cuda.Event(enable_timing=True, external=True)
    ev['s1'] = torch.cuda.Event(enable_timing=True, external=True)
    gr = torch.cuda.CUDAGraph()
    ev['on'] = True
    with torch.cuda.graph(gr):
        ev['s0'].record(); step(); ev['s1'].record()
    ev['on'] = False
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    ks, ss, per = [], [], []
    for grp in range(6):
        t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / 10 * 1e3)
        ks.append(ev['e0'].elapsed_time(ev['e1'])); ss.append(ev['s0'].elapsed_time(ev['s1']))
    res['in_graph_call_ms'] = ks; res['in_graph_step_ms'] = ss; res['period_ms'] = per
except Exception as exc:
    res['in_graph_error'] = f'{type(exc).__name__}: {exc}'
    ev['on'] = False
    torch.cuda.synchronize()
# plain graph (no event nodes): period
gr2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr2):
    step()
for _ in range(5): gr2.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): gr2.replay()
torch.cuda.synchronize()
res['plain_graph_period_ms'] = (time.perf_counter() - t0) / 20 * 1e3
# bare back-to-back eager launches of the native call
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
native(*last['a'], **last['k'])
e0.record()
for _ in range(10): native(*last['a'], **last['k'])
e1.record(); torch.cuda.synchronize()
res['bare_back_to_back_ms'] = e0.elapsed_time(e1) / 10
# the host + front-end cost of replaying a graph at all: one 1-element kernel
xx = torch.zeros(1, device=dev)
gr3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr3):
    xx.add_(1.0)
for _ in range(5): gr3.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): gr3.replay()
torch.cuda.synchronize()
res['one_tiny_kernel_graph_period_ms'] = (time.perf_counter() - t0) / 200 * 1e3
print(res)
