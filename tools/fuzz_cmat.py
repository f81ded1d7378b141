#!/usr/bin/env python3
"""Random shapes through the conditional posterior: matrix-pipe passes (vibo_cmean.hip) against the VALU passes
(VIBO_FLAG_COND_VALU) on the same inputs -- item counts with every kind of tail, row strides that are / are not 16-byte
multiples, rows through row_index, fp32 rows and cell codes, both row-split kernels, forward-only calls.
   python tools/fuzz_cmat.py [--cases 60] [--seed 0]"""
import argparse, os, random, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

ap = argparse.ArgumentParser()
ap.add_argument('--cases', type=int, default=60)
ap.add_argument('--seed', type=int, default=0)
a = ap.parse_args()
rnd = random.Random(a.seed)
d = torch.device('cuda:0')
bad = 0
for case in range(a.cases):
    irt = rnd.choice([1, 2, 2, 3])
    A = rnd.choice([1, 1, 2, 3, 4, 5, 8])
    I = rnd.choice([rnd.randint(4, 70), rnd.randint(60, 260), rnd.randint(250, 1100), rnd.randint(1000, 2600)])
    B = rnd.choice([1, 16, 77, 300, 4096, 4097, 4160, 5000, 6001])
    codes = rnd.random() < 0.6
    gather = rnd.random() < 0.4
    flows = rnd.choice([0, 0, 2])
    drop = rnd.random() < 0.3
    grad = rnd.random() < 0.8
    pad = rnd.choice(['pack', 'tight4', 'odd'])          # code-row stride: pack_cell_codes' / I rounded to 4 / that + 4
    kern = rnd.choice([0, _lib.FLAG_KERNEL_MATRIX, _lib.FLAG_KERNEL_VALU])
    g = torch.Generator(device=d).manual_seed(1000 * a.seed + case)
    P = B + (37 if gather else 0)
    resp = (torch.rand(P, I, device=d, generator=g) < 0.55).float()
    mask = torch.rand(P, I, device=d, generator=g) >= rnd.choice([0.0, 0.15, 0.6])
    if drop:
        mask[:, 0] = True
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=True, n_flows=flows, drop_missing=drop)
    table = torch.randn(2, I, 2 * A, device=d, generator=g) * 0.6
    item = torch.randn(I, spec.item_dim, device=d, generator=g) * 0.8
    eps = torch.randn(B, A, device=d, generator=g)
    flow = torch.randn(flows, 2 * A + 1, device=d, generator=g) * 0.5 if flows else None
    rows = torch.randperm(P, device=d, generator=g)[:B].contiguous() if gather else None
    if codes:
        cc = ops.pack_cell_codes(resp, mask).codes
        if pad != 'pack':
            st = (I + 3) // 4 * 4 + (4 if pad == 'odd' else 0)
            buf = torch.full((P, st), 2, dtype=torch.uint8, device=d)
            buf[:, :I] = cc
            cc = buf[:, :I]
        r = m = cc
        code = _lib.MASK_CODES
    else:
        r_, m_ = ops.pad_rows(resp, mask)
        r, m, code = ops.prepare_rows(r_, m_)
    reg = _lib.REG_SAMPLED if flows else _lib.REG_KL
    out = {}
    for name, fl in (('mfma', _lib.FLAG_COND_MATRIX), ('valu', _lib.FLAG_COND_VALU)):
        ops.DESC_FLAGS = kern | fl
        out[name] = ops._hip_launch_elbo(spec, r, m, code, rows, table, item, eps, flow, reg, grad, B)
    torch.cuda.synchronize()
    x, y = out['mfma'], out['valu']
    errs = {'mu': float((x.ability_mu - y.ability_mu).abs().max() / max(1.0, float(y.ability_mu.abs().max()))),
            'lv': float((x.ability_logvar - y.ability_logvar).abs().max() / max(1.0, float(y.ability_logvar.abs().max()))),
            'll': abs(float(x.scalars[0]) - float(y.scalars[0])) / max(1.0, abs(float(y.scalars[0])))}
    if grad:
        for s_ in range(2):
            gx, gy = x.grad_table(s_), y.grad_table(s_)
            errs[f'gt{s_}'] = float((gx - gy).abs().max() / max(1e-6, float(gy.abs().max())))
        gx, gy = x.grad_item((I, spec.item_dim)), y.grad_item((I, spec.item_dim))
        errs['gi'] = float((gx - gy).abs().max() / max(1e-6, float(gy.abs().max())))
    worst = max(errs.values())
    ok = worst < 5e-5 and all(v == v for v in errs.values())
    bad += not ok
    if not ok and grad:
        # which of the two forms is off: both against the fp64 oracle (test infrastructure)
        from oracle import vibo_table_ref as T
        rs, ms = (resp[rows], mask[rows]) if gather else (resp, mask)
        flows_ = [(f[:A].double().cpu(), f[A:2 * A].double().cpu(), f[2 * A:].double().cpu()) for f in flow] if flows else None
        ref = T.fused_elbo_ref(table.cpu().double(), item.cpu().double(), rs.cpu().double(), ms.cpu(), eps.cpu().double(), irt_model=irt,
                               ability_dim=A, conditional_posterior=True, replace_missing_with_prior=not drop,
                               mode='sampled' if flows else 'kl', flow_uhat_w_b=flows_)
        for name in ('mfma', 'valu'):
            for s_ in range(2):
                if 'g_table' in ref:
                    gy = ref['g_table'][s_].float().reshape(out[name].grad_table(s_).shape)
                    print(f'   {name} set {s_} vs the fp64 oracle: {float((out[name].grad_table(s_).cpu() - gy).abs().max() / max(1e-6, float(gy.abs().max()))):.1e}')
    print(f'{"ok " if ok else "BAD"} irt={irt} A={A} B={B} I={I} codes={codes} pad={pad} gather={gather} flows={flows} drop={drop} grad={grad} kern={kern}: '
          + ' '.join(f'{k}={v:.1e}' for k, v in errs.items()))
print(f'# {bad} of {a.cases} cases disagree')
sys.exit(1 if bad else 0)
