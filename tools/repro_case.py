#!/usr/bin/env python3
"""One configuration of tools/fuzz_parity.py, for debugging: python tools/repro_case.py irt A B I cond flows drop missing pad seed"""
import os
import sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd')):
    sys.path.insert(0, p)
import torch
from oracle import vibo_oracle as O
from oracle import vibo_table_ref as T
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

irt, A, B, I, cond, n_flows, drop, missing, pad, seed = sys.argv[1:11]
irt, A, B, I, n_flows, seed = int(irt), int(A), int(B), int(I), int(n_flows), int(seed)
cond, drop, pad, missing = cond == '1', drop == '1', pad == '1', float(missing)
d = torch.device('cuda:0')
spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=cond, drop_missing=drop, n_flows=n_flows)
g = torch.Generator().manual_seed(seed)
resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=missing)
D = O.item_feat_dim(irt, A)
table = torch.randn((2, I, 2 * A) if cond else (2, 2 * A), generator=g) * 0.7
item = torch.randn(I, D, generator=g) * 0.6
eps = torch.randn(B, A, generator=g)
flow = None
if n_flows:
    raw_f = torch.randn(n_flows, 2 * A + 1, generator=g) * 0.7
    flow = torch.stack([torch.cat([T.flow_uhat(f[:A], f[A:2 * A]), f[A:]]) for f in raw_f])
flows = [(f[:A].double(), f[A:2 * A].double(), f[2 * A:].double()) for f in flow] if n_flows else None
ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(), irt_model=irt, ability_dim=A,
                       conditional_posterior=cond, replace_missing_with_prior=not drop, mode='sampled' if n_flows else 'kl',
                       flow_uhat_w_b=flows)
r_, m_ = (ops.pad_rows(resp.to(d), mask.bool().to(d)) if pad else (resp.to(d), mask.bool().to(d)))
r = ops.prepare_response(r_)
m, code = ops.prepare_mask(m_)
raw = ops._hip_launch_elbo(spec, r, m, code, None, table.to(d).contiguous(), item.to(d).contiguous(), eps.to(d).contiguous(),
                           flow.to(d).contiguous() if flow is not None else None,
                           _lib.REG_SAMPLED if n_flows else _lib.REG_KL, True, B)
torch.cuda.synchronize()
gi, gr = raw.grad_item((I, D)).cpu(), ref['g_item'].float()
err = (gi - gr).abs()
print('args', sys.argv[1:], 'll', float(raw.scalars[0]), float(ref['ll']), 'max g_item err', float(err.max()), 'at item', int(err.max(1).values.argmax()),
      'ref max', float(gr.abs().max()))
print('per-item err (first 8 / last 8):', err.max(1).values[:8].tolist(), err.max(1).values[-8:].tolist())
for s_ in range(2):
    print('g_table', s_, float((raw.grad_table(s_).cpu() - ref['g_table'][s_].float()).abs().max()), float(ref['g_table'][s_].abs().max()))
