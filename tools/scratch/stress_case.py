import sys, torch
sys.path.insert(0, '/root/repo/variational-item-response-theory-public_amd'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec
from oracle import vibo_oracle as O
dev = torch.device('cuda:0')
irt, A, B, I, cond, n_flows, drop, gather = 3, 4, 64, 1500, True, 0, False, True
g = torch.Generator().manual_seed(0 + 7 * I + A)
resp, mask = O.simulate_responses(irt, B + 6, I, A, generator=g, missing_frac=0.2)
spec = ElboSpec(irt_model=irt, ability_dim=A, n_flows=n_flows, conditional=cond, drop_missing=drop)
table = (torch.randn(*spec.table_shape(I), generator=g) * 0.6).to(dev)
item = (torch.randn(I, spec.item_dim, generator=g) * 0.7).to(dev)
eps = torch.randn(B + 6, A, generator=g).to(dev)[:B].contiguous()
resp, mask = resp.to(dev), mask.bool().to(dev)
rows = torch.randperm(B + 6)[:B].to(dev)
r_, m_ = ops.pad_rows(resp, mask)
r, m, code = ops.prepare_rows(r_, m_)
c, cm, ccode = ops.prepare_rows(ops.pack_cell_codes(resp, mask), None)
for flags in (_lib.FLAG_KERNEL_MATRIX | _lib.FLAG_NO_EMIT_CODES | _lib.FLAG_COND_VALU, _lib.FLAG_KERNEL_VALU | _lib.FLAG_COND_VALU, _lib.FLAG_KERNEL_MATRIX | _lib.FLAG_COND_MATRIX):
    ops.DESC_FLAGS = flags
    for name, args in (('fp32', (r, m, code)), ('codes', (c, cm, ccode))):
        base = None; nbad = 0
        for it in range(300):
            # churn the allocator / workspace contents between calls
            junk = torch.randn(1 << 20, device=dev) * float('nan') if it % 7 == 0 else None
            o = ops._hip_launch_elbo(spec, args[0], args[1], args[2], rows, table, item, eps, None, _lib.REG_KL, True, B)
            f = o.flat.clone()
            if base is None: base = f
            elif not torch.equal(f, base):
                nbad += 1
                if nbad == 1:
                    d = (f != base).nonzero().flatten()
                    print('   first mismatch at iter', it, 'n elems', d.numel(), 'idx', d[:8].tolist(), 'vals', f[d[:3]].tolist(), base[d[:3]].tolist())
        print(f'flags={flags} {name}: {nbad} of 299 repeats differ from the first')
print('--- random row subsets, fp32 rows vs cell codes')
for flags in (_lib.FLAG_KERNEL_MATRIX | _lib.FLAG_NO_EMIT_CODES | _lib.FLAG_COND_VALU, _lib.FLAG_KERNEL_VALU | _lib.FLAG_COND_VALU, _lib.FLAG_KERNEL_MATRIX | _lib.FLAG_COND_VALU):
    ops.DESC_FLAGS = flags
    nbad = 0
    for it in range(400):
        rows = torch.randperm(B + 6)[:B].to(dev)
        a = ops._hip_launch_elbo(spec, r, m, code, rows, table, item, eps, None, _lib.REG_KL, True, B)
        b = ops._hip_launch_elbo(spec, c, cm, ccode, rows, table, item, eps, None, _lib.REG_KL, True, B)
        if not torch.equal(a.flat, b.flat):
            nbad += 1
            if nbad <= 2:
                d = (a.flat != b.flat).nonzero().flatten()
                print('   mismatch iter', it, 'n elems', d.numel(), 'first idx', d[:6].tolist(), a.flat[d[:3]].tolist(), b.flat[d[:3]].tolist(),
                      'mu equal', bool(torch.equal(a.ability_mu, b.ability_mu)), 'rows', rows[:8].tolist())
    print(f'flags={flags}: {nbad} of 400 row subsets differ between the formats')
