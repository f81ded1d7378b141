#!/usr/bin/env python3
"""Launch the per-term MLP decoder kernel (vibo_decoder_fwd_bwd) on a synthetic problem (for rocprofv3 / timing).
   python tools/profile_decoder.py [--persons B] [--items I] [--mode deep|residual|link] [--no-grad] [--iters N]"""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from vibo_amd import decoder as D

ap = argparse.ArgumentParser()
ap.add_argument('--persons', type=int, default=100_000)
ap.add_argument('--items', type=int, default=1000)
ap.add_argument('--mode', default='deep')
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--no-grad', action='store_true')
ap.add_argument('--torch-compare', type=int, default=0, metavar='B', help='also time the same network as plain PyTorch fp32 ops (rocBLAS GEMMs + autograd) on B persons')
a = ap.parse_args()
d = torch.device('cuda:0')
g = torch.Generator(device=d).manual_seed(0)
B, I, H = a.persons, a.items, 64
rn = lambda *s, sc=1.0: torch.randn(*s, device=d, generator=g) * sc
resp = (torch.rand(B, I, device=d, generator=g) < 0.5).float()
mask = torch.rand(B, I, device=d, generator=g) >= 0.1
args = [resp, mask.view(torch.uint8), rn(I, H, sc=0.7) if a.mode != 'link' else None, rn(B, H, sc=0.7),
        rn(B, I, sc=1.5) if a.mode != 'deep' else None, None, rn(H, sc=0.5) if a.mode == 'link' else None,
        rn(H, H, sc=0.18), rn(H, sc=0.1), rn(H, sc=0.25), rn(1, sc=0.1), 1.0 if a.mode == 'residual' else 0.0, not a.no_grad]
for _ in range(2):
    out = D._launch(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    out = D._launch(*args)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
terms = B * I
flop = terms * (3 if not a.no_grad else 1) * 2 * H * H          # algorithmic: one 64x64 mat-vec forward, two backward
print(f'mode={a.mode} B={B} I={I} grad={not a.no_grad}: {ms:.3f} ms/call, {terms / ms / 1e6:.2f} G terms/s, '
      f'{flop / ms / 1e9:.1f} TFLOP/s algorithmic (x3 f16 MFMA passes issued), ll={float(out["ll_part"].sum()):.1f}')

if a.torch_compare:
    import torch.nn.functional as F
    Bc = a.torch_compare
    U, V, W2, b2, w3, b3 = [t.clone().requires_grad_(True) if t is not None else None for t in (args[2], args[3][:Bc], args[7], args[8], args[9], args[10])]
    r, m = resp[:Bc], mask[:Bc]

    def step():
        z1 = V.unsqueeze(1) + (U.unsqueeze(0) if U is not None else 0.0)
        o = F.elu(F.elu(z1) @ W2.t() + b2) @ w3 + b3
        ll = -(F.binary_cross_entropy_with_logits(o, r, reduction='none') * m).sum()
        ll.backward()
        return ll
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f'plain PyTorch fp32 (materialised [B,I,64] activations, rocBLAS + autograd) B={Bc}: {ms:.3f} ms/call, '
          f'{Bc * I / ms / 1e6:.2f} G terms/s, {Bc * I * 3 * 2 * H * H / ms / 1e9:.1f} TFLOP/s')
