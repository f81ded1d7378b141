import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec
from oracle import vibo_oracle as O
d = torch.device('cuda:0')
ops.DESC_FLAGS = _lib.FLAG_KERNEL_MATRIX
irt, A, B, I = 2, 2, 64, 256
g = torch.Generator().manual_seed(5)
resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=0.1)
table = torch.randn(2, 2 * A, generator=g) * 0.7
item = torch.randn(I, 3, generator=g)
eps = torch.randn(B, A, generator=g)
spec = ElboSpec(irt_model=irt, ability_dim=A)
print('plan', ops.plan_kernel(spec, B, I))
for val, col in ((3e9, A), (1e12, A), (float('inf'), A), (float('nan'), A), (3e9, 0), (1e20, 0)):
    it = item.clone(); it[3, col] = val
    r = ops.prepare_response(resp.to(d)); m, code = ops.prepare_mask(mask.to(d).bool())
    raw = ops._hip_launch_elbo(spec, r, m, code, None, table.to(d), it.to(d).contiguous(), eps.to(d), None, _lib.REG_KL, True, B)
    torch.cuda.synchronize()
    print(val, col, 'LL', float(raw.scalars[0]), 'g_item nan', bool(torch.isnan(raw.grad_item((I, 3))).any()))
