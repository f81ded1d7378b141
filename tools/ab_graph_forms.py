#!/usr/bin/env python3
"""One hipGraph per train step against two (fused ELBO call | epilogue) replayed back to back: step period of both forms,
interleaved, on the bench workload.   python tools/ab_graph_forms.py [--persons P] [--dist]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')]
import torch
from vibo_amd.torch_core.models import VIBO_2PL
from vibo_amd.trainer import FusedTrainer
ap = argparse.ArgumentParser()
ap.add_argument('--persons', type=int, default=1_000_000)
ap.add_argument('--items', type=int, default=1000)
ap.add_argument('--ability-dim', type=int, default=8)
ap.add_argument('--dist', action='store_true', help='1-rank nccl group, all-reduce captured')
ap.add_argument('--steps', type=int, default=40)
a = ap.parse_args()
dev = torch.device('cuda:0')
dist = None
if a.dist:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
P, I, A = a.persons, a.items, a.ability_dim
g = torch.Generator(device=dev).manual_seed(1)
resp = (torch.rand(P, I, device=dev, generator=g) < 0.5).float()
mask = torch.rand(P, I, device=dev, generator=g) >= 0.1
torch.manual_seed(0)
model = VIBO_2PL(A, I, ability_merge='product').to(dev)
if dist is not None:
    model.enable_person_sharding(lambda flat: dist.all_reduce(flat), seed=0, rank=0)
tr = FusedTrainer(model, lr=5e-3, rng='native', seed=0)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): tr.step(resp, mask)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    tr.step(resp, mask)
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(ga, pool=g1.pool()):
    raw = tr.forward_backward(resp, mask)
with torch.cuda.graph(gb, pool=g1.pool()):
    if dist is not None: dist.all_reduce(raw.flat)
    tr.update()
def one():
    g1.replay()
def two():
    ga.replay(); gb.replay()
def two_ev():
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ga.replay(); e1.record(); gb.replay()
# more forms, to see what about the graph boundary matters (box-dependent: DESIGN 3.1a)
tiny = torch.zeros(64, device=dev)
gm = torch.cuda.CUDAGraph()
with torch.cuda.graph(gm, pool=g1.pool()):          # one graph with a tiny fill between the two kernels
    raw_m = tr.forward_backward(resp, mask)
    tiny.add_(1.0)
    tr.update()
gt = torch.cuda.CUDAGraph()
with torch.cuda.graph(gt, pool=g1.pool()):          # a graph of its own that does next to nothing
    tiny.add_(1.0)
def one_fill():
    gm.replay()
def three():
    ga.replay(); gt.replay(); gb.replay()
def one_then_tiny():
    g1.replay(); gt.replay()
def eager():
    tr.step(resp, mask)
forms = (('one', one), ('two', two), ('two_ev', two_ev), ('one_fill', one_fill), ('three', three), ('one_then_tiny', one_then_tiny), ('eager', eager))
res = {k: [] for k, _ in forms}
for rep in range(5):
    for name, fn in forms:
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps): fn()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
print({k: [round(x, 4) for x in v] for k, v in res.items()})
if dist is not None: dist.destroy_process_group()
