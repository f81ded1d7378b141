import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'variational-item-response-theory-public_amd'))
import test_gpu_cell_codes as T
from vibo_amd import ops, _lib
from vibo_amd.ops import ElboSpec
dev = T.dev
for (irt, A, B, I, cond, n_flows, drop, gather) in [(3, 8, 64, 1028, False, 2, False, True), (3, 8, 64, 1028, False, 2, False, False), (3, 8, 64, 1000, False, 2, False, True), (2, 8, 64, 1028, False, 2, False, True), (3, 8, 64, 1028, False, 0, False, True)]:
  for want_grad in (True,):
    spec, resp, mask, table, item, eps, flow = T.problem(irt, A, B + 6, I, cond, n_flows)
    spec = ElboSpec(irt_model=irt, ability_dim=A, n_flows=n_flows, conditional=cond, drop_missing=drop)
    rows = torch.randperm(B + 6, generator=torch.Generator().manual_seed(B + I))[:B].to(dev) if gather else None
    if rows is None: resp, mask = resp[:B], mask[:B]
    reg = _lib.REG_SAMPLED if n_flows else _lib.REG_KL
    r_, m_ = ops.pad_rows(resp, mask)
    r, m, code = ops.prepare_rows(r_, m_)
    ref = ops._hip_launch_elbo(spec, r, m, code, rows, table, item, eps, flow, reg, want_grad, B)
    c, cm, ccode = ops.prepare_rows(ops.pack_cell_codes(resp, mask), None)
    got = ops._hip_launch_elbo(spec, c, cm, ccode, rows, table, item, eps, flow, reg, want_grad, B)
    n = got.flat.numel()
    d = (got.flat[:n] != ref.flat[:n]).nonzero().flatten().tolist()
    print((irt, A, B, I, n_flows, gather), 'n', n, 'NUM_SCALARS', _lib.NUM_SCALARS, 'diff idx', d[:40], len(d))
    for i in d[:8]: print('   ', i, got.flat[i].item(), ref.flat[i].item())
    for nm in ('ability_mu', 'ability_logvar', 'ability', 'ability_k', 'ability_ladj'):
        x, y = getattr(got, nm, None), getattr(ref, nm, None)
        if x is not None and y is not None and x.numel(): print('   ', nm, int((x != y).sum()))
