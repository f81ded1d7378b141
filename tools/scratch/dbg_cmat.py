import sys, torch
sys.path.insert(0, '/root/repo/variational-item-response-theory-public_amd'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec
from oracle import vibo_oracle as O
d = torch.device('cuda:0')
def case(irt, A, B, I, n_flows, drop, codes, flags):
    g = torch.Generator().manual_seed(A * I + B)
    resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=0.2)
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=True, n_flows=n_flows, drop_missing=drop)
    table = torch.randn(2, I, 2 * A, generator=g) * 0.7
    item = torch.randn(I, spec.item_dim, generator=g)
    eps = torch.randn(B, A, generator=g)
    flow = (torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5).to(d) if n_flows else None
    if drop:
        mask[:, 0] = 1; resp[:, 0] = resp[:, 0].clamp(min=0)
    if codes:
        r = m = ops.pack_cell_codes(resp.to(d), mask.to(d)).codes; code = _lib.MASK_CODES
    else:
        r = ops.prepare_response(resp.to(d)); m, code = ops.prepare_mask(mask.bool().to(d))
    ops.DESC_FLAGS = flags
    outs = []
    for _ in range(3):
        o = ops._hip_launch_elbo(spec, r, m, code, None, table.to(d), item.to(d), eps.to(d), flow, _lib.REG_SAMPLED if n_flows else _lib.REG_KL, True, B)
        torch.cuda.synchronize()
        outs.append((o.flat.clone(), o.ability_mu.clone(), o.ability_logvar.clone(), o.grad_table(0).clone()))
    a = outs[0]
    for b in outs[1:]:
        print('   flat', bool(torch.equal(a[0], b[0])), 'mu', bool(torch.equal(a[1], b[1])), 'lv', bool(torch.equal(a[2], b[2])), 'gtab', bool(torch.equal(a[3], b[3])),
              'n mu diff', int((a[1] != b[1]).sum()), 'max', float((a[1] - b[1]).abs().max()))
for args in [(3, 3, 4500, 95, 2, True, False), (3, 3, 4500, 95, 0, True, False), (2, 3, 4500, 95, 0, False, False), (2, 3, 4500, 95, 0, False, True),
             (2, 3, 4500, 128, 0, False, False), (2, 3, 4500, 96, 0, False, False)]:
    for fl in (0, _lib.FLAG_COND_VALU):
        print(args, 'flags', fl)
        case(*args, fl)
