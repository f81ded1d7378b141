#!/usr/bin/env python3
"""Decoder kernel scratch check: the same vibo_decoder_fwd_bwd call with every output / scratch buffer pre-filled with 0x00 and
with 0xFF bytes must return identical results (nothing read that the call did not write; outputs fully overwritten).
   python tools/decoder_guard_check.py      (needs the MI355X; test infrastructure)"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/variational-item-response-theory-public_amd')
import torch
from vibo_amd import decoder as D
dev = torch.device('cuda:0')
fill = [0]
real_empty = torch.empty
class T:
    def __getattr__(self, k): return getattr(torch, k)
    @staticmethod
    def empty(*shape, **kw):
        t = real_empty(*shape, **kw)
        if t.numel(): t.view(torch.uint8).fill_(fill[0])
        return t
D.torch = T()
for (mode, B, I) in [('deep', 64, 700), ('deep', 64, 1500), ('link', 33, 95), ('residual', 257, 130), ('deep', 1, 1)]:
    g = torch.Generator(device=dev).manual_seed(B + I)
    rn = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    H = 64
    resp = (torch.rand(B, I, device=dev, generator=g) < 0.5).float()
    mask = (torch.rand(B, I, device=dev, generator=g) >= 0.1).view(torch.uint8)
    args = [resp, mask, rn(I, H, sc=0.7) if mode != 'link' else None, rn(B, H, sc=0.7), rn(B, I) if mode != 'deep' else None, None,
            rn(H, sc=0.5) if mode == 'link' else None, rn(H, H, sc=0.18), rn(H, sc=0.1), rn(H, sc=0.25), rn(1, sc=0.1), 1.0 if mode == 'residual' else 0.0, True]
    outs = []
    for f in (0, 255):
        fill[0] = f
        o = D._launch(*args)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in o.items()})
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        same = torch.equal(a, b)
        if not same:
            print(mode, B, I, k, 'DIFFERS', int((a != b).sum()), 'of', a.numel(), 'nan in b', int(b.isnan().sum()))
    print(mode, B, I, 'checked')
