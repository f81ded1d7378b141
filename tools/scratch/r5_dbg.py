import sys, os
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec
import test_gpu_cell_codes as T
dev = torch.device('cuda:0')
ops.DESC_FLAGS = _lib.FLAG_KERNEL_MATRIX | _lib.FLAG_COND_MATRIX
def run(irt, A, B, I, cond, n_flows, drop, gather, want_grad=True):
    spec, resp, mask, table, item, eps, flow = T.problem(irt, A, B + 6, I, cond, n_flows)
    spec = ElboSpec(irt_model=irt, ability_dim=A, n_flows=n_flows, conditional=cond, drop_missing=drop)
    rows = torch.randperm(B + 6, generator=torch.Generator().manual_seed(B + I))[:B].to(dev) if gather else None
    if rows is None:
        resp, mask = resp[:B], mask[:B]
    reg = _lib.REG_SAMPLED if n_flows else _lib.REG_KL
    r_, m_ = ops.pad_rows(resp, mask)
    r, m, code = ops.prepare_rows(r_, m_)
    outs = []
    for rep in range(2):
        ref = ops._hip_launch_elbo(spec, r, m, code, rows, table, item, eps, flow, reg, want_grad, B)
        c, cm, ccode = ops.prepare_rows(ops.pack_cell_codes(resp, mask), None)
        got = ops._hip_launch_elbo(spec, c, cm, ccode, rows, table, item, eps, flow, reg, want_grad, B)
        outs.append((ref.flat.clone(), got.flat.clone()))
    x, y = outs[0]
    d = (x != y).nonzero().flatten().tolist()
    print((irt, A, B, I, cond, n_flows, drop, gather), 'n_diff', len(d), 'first', d[:12], 'of', x.numel(),
          'ref rep-stable', bool(torch.equal(outs[0][0], outs[1][0])), 'codes rep-stable', bool(torch.equal(outs[0][1], outs[1][1])))
    for k in d[:6]:
        print('   idx', k, float(x[k]), float(y[k]))
    if d:
        # third opinion: the VALU row-split kernel on the fp32 rows
        ops.DESC_FLAGS = _lib.FLAG_KERNEL_VALU
        val = ops._hip_launch_elbo(spec, r, m, code, rows, table, item, eps, flow, reg, want_grad, B).flat.clone()
        ops.DESC_FLAGS = _lib.FLAG_KERNEL_MATRIX | _lib.FLAG_COND_MATRIX
        D = spec.item_dim
        off_item = x.numel() - I * D
        print('   off_item', off_item, 'D', D)
        dd = torch.tensor(d)
        it = dd[dd >= off_item] - off_item
        print('   item-grad diffs by column d:', torch.bincount(it % D, minlength=D).tolist(), ' head diffs', dd[dd < off_item].tolist())
        print('   item diffs raw (idx - off_item):', it[:48].tolist())
        bad = ((x - val).abs() / (val.abs() + 1e-3) > 1e-3).nonzero().flatten() - off_item
        print('   BAD vs VALU (idx - off_item):', bad[:64].tolist())
        ex = (x - val).abs() / (val.abs() + 1e-3); ey = (y - val).abs() / (val.abs() + 1e-3)
        print('   max rel dev from VALU kernel: fp32-matrix', float(ex.max()), 'at', int(ex.argmax()), ' codes-matrix', float(ey.max()), 'at', int(ey.argmax()))
        print('   fp32-matrix bad entries', int((ex > 1e-3).sum()), ' codes-matrix bad entries', int((ey > 1e-3).sum()))
for c in [(3, 8, 64, 1028, False, 2, False, True), (3, 8, 64, 2048, False, 2, False, True), (3, 8, 64, 1000, False, 2, False, True), (3, 1, 64, 2048, False, 2, False, True), (3, 4, 64, 2048, False, 1, False, True)]:
    run(*c)
