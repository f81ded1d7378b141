import os, sys, copy
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from vibo_amd import _lib, ops
from vibo_amd.torch_core.models import VIBO_2PL, VIBO_3PL
from vibo_amd.trainer import FusedTrainer
from oracle import vibo_oracle as O
dev = torch.device('cuda:0')
def up(n): return (n + 63) & ~63
def run(cls, A, I, B, kw, seed=3, fscale=1.0, perturb=0.0):
    g = torch.Generator().manual_seed(A * 100 + I)
    resp, mask = O.simulate_responses(cls.IRT, B, I, A, generator=g, missing_frac=0.15)
    resp, mask = resp.to(dev), mask.bool().to(dev)
    torch.manual_seed(seed)
    ref = cls(A, I, ability_merge='product', **kw).to(dev)
    if fscale != 1.0:
        with torch.no_grad():
            for st in (ref.ability_norm_flows, ref.item_norm_flows):
                for fl in st.flows:
                    fl.u.mul_(fscale); fl.w.mul_(fscale)
    fus = copy.deepcopy(ref)
    if perturb:
        with torch.no_grad():
            fus.ability_encoder.mlp[4].bias.mul_(1.0 + perturb)
    tr = FusedTrainer(fus, lr=5e-3)
    torch.manual_seed(100)
    loss_ref = ref.elbo_step(resp, mask, annealing_factor=1.0)
    loss_ref.backward()
    torch.manual_seed(100)
    loss_f = tr.step(resp, mask, beta=1.0)
    torch.cuda.synchronize()
    H, D, F = 64, ref.item_feat_dim, ref.n_norm_flows
    cond = ref.conditional_posterior
    xin = 1 + (D if cond else 0); rows = 2 * (I if cond else 1); O_ = 2 * A
    n_mlp = H * xin + H + H * H + H + O_ * H + O_
    n_rb = (rows + 63) // 64; n_ib = (I + 255) // 256
    s_pack = 0; s_tanh = up(8 * 22); s_parts = s_tanh + up(I * max(F, 1)); s_gx = s_parts + up(n_ib * 4)
    s_mrec = s_gx + up(rows * 10); s_frec = s_mrec + up(n_rb * n_mlp)
    mine = tr.scratch[s_mrec:s_mrec + n_rb * n_mlp].view(n_rb, n_mlp).double().sum(0).float()
    mlp = ref.ability_encoder.mlp
    theirs = torch.cat([p.grad.reshape(-1) for p in (mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, mlp[4].weight, mlp[4].bias)])
    names = [('w0', H * xin), ('b0', H), ('w1', H * H), ('b1', H), ('w2', O_ * H), ('b2', O_)]
    print(cls.__name__, A, I, kw, 'loss', float(loss_ref), float(loss_f))
    o = 0
    for nm, n in names:
        a, b = mine[o:o + n], theirs[o:o + n]
        err = (a - b).abs()
        k = int(err.argmax())
        print(f'  {nm}: max|g| {float(b.abs().max()):.4e}  max err {float(err.max()):.3e} at {k} (theirs {float(b[k]):.4e} mine {float(a[k]):.4e})  sign flips {int(((a * b) < 0).sum())}')
        o += n
    # item-side gradients through the first Adam step: sign only
    for nm, pr, pf in (('item_mu', ref.item_encoder.mu_lookup.weight, fus.item_encoder.mu_lookup.weight), ('item_lv', ref.item_encoder.logvar_lookup.weight, fus.item_encoder.logvar_lookup.weight)):
        moved = (pf.detach() - pr.detach())           # fused moved by -lr sign(g_mine); ref not stepped yet
        gs = pr.grad
        flips = ((-moved * gs) < 0) & (gs.abs() > 1e-3 * gs.abs().max())
        print(f'  {nm}: flips among big grads {int(flips.sum())} of {int((gs.abs() > 1e-3 * gs.abs().max()).sum())}; cols {flips.nonzero()[:8].tolist()}')
    if F:
        for st, nm in ((ref.ability_norm_flows, 'ab'), (ref.item_norm_flows, 'it')):
            fst = fus.ability_norm_flows if nm == 'ab' else fus.item_norm_flows
            for k, (fr, ff) in enumerate(zip(st.flows, fst.flows)):
                for pn in ('u', 'w', 'b'):
                    pr, pf = getattr(fr, pn), getattr(ff, pn)
                    moved = pf.detach() - pr.detach()
                    bad = ((-moved * pr.grad) < 0)
                    if bad.any():
                        print(f'  flow {nm}[{k}].{pn}: sign flips at {bad.nonzero().flatten().tolist()} grads {pr.grad[bad].tolist()}')
print('== seed 3'); run(VIBO_3PL, 8, 200, 48, dict(conditional_posterior=True, n_norm_flows=2))
print('== seed 4'); run(VIBO_3PL, 8, 200, 48, dict(conditional_posterior=True, n_norm_flows=2), seed=4)
print('== seed 5'); run(VIBO_3PL, 8, 200, 48, dict(conditional_posterior=True, n_norm_flows=2), seed=5)
print('== seed 3, flow params x 0.5'); run(VIBO_3PL, 8, 200, 48, dict(conditional_posterior=True, n_norm_flows=2), fscale=0.5)
