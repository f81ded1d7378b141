"""Pin the CPU oracle (oracle/vibo_oracle.py) against every golden vector that
tools/gen_golden.py produced from the real reference."""
import os

import numpy as np
import torch

from conftest import GOLDEN_DIR, rel_err
from oracle import vibo_oracle as O

# fp32 tolerances: forward <= 1e-4 rel (observed ~1e-6),
# posteriors <= 1e-5 abs, grads <= 1e-4 of the tensor's max-abs.
TOL_LOSS = 2e-5
TOL_POST = 2e-5
TOL_GRAD = 2e-4


def test_forward_matches_reference(golden):
    out = O.elbo_forward(golden.sd, golden.response, golden.mask, golden.eps_item,
                         golden.eps_ability, **golden.cfg)
    assert rel_err(out['loss'], golden.out['loss']) < TOL_LOSS
    for k in ('ability_mu', 'ability_logvar', 'ability', 'item_feat'):
        assert (out[k] - golden.out[k]).abs().max() < TOL_POST * max(1.0, float(golden.out[k].abs().max())), k
    assert (out['response_mu'] - golden.out['response_mu']).abs().max() < 1e-5
    if golden.meta['n_norm_flows'] > 0:
        for k in ('ability_k', 'ability_logabsdetjac', 'item_feat_k', 'item_feat_logabsdetjac'):
            assert (out[k] - golden.out[k]).abs().max() < 1e-4, k


def test_grads_match_reference(golden):
    _, grads = O.elbo_loss_and_grads(golden.sd, golden.response, golden.mask, golden.eps_item,
                                     golden.eps_ability, **golden.cfg)
    for k, g_ref in golden.grad.items():
        scale = float(g_ref.abs().max())
        if scale == 0.0:
            assert float(grads[k].abs().max()) < 1e-6, k
            continue
        assert rel_err(grads[k], g_ref) < TOL_GRAD, k


def test_adam_trajectory_matches_reference(golden):
    """3 Adam(lr=5e-3) steps on the same batch/eps reproduce the reference's
    parameters (vibo.py:221,243-268)."""
    params = {k: v.clone().requires_grad_(True) for k, v in golden.sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=5e-3)
    for step in range(3):
        opt.zero_grad()
        out = O.elbo_forward(params, golden.response, golden.mask, golden.eps_item,
                             golden.eps_ability, **golden.cfg)
        out['loss'].backward()
        opt.step()
        ref = golden.adam1 if step == 0 else golden.adam3 if step == 2 else None
        if ref is not None:
            for k, v in ref.items():
                # Adam's first steps are sign-like (lr-sized), so compare absolutely
                assert (params[k].detach() - v).abs().max() < 2e-4, (step, k)


def test_saturation_semantics():
    """The probability clamp of torch.distributions.Bernoulli (utils.py:46-49):
    ll floors at log(eps32); gradient is exactly zero outside
    [-15.942385, 16.635532]."""
    z = np.load(os.path.join(GOLDEN_DIR, 'saturation.npz'))
    l = torch.from_numpy(z['logit'])
    for x in (0, 1):
        lv = l.clone().requires_grad_(True)
        ll = O.masked_bernoulli_ll(torch.full_like(lv, float(x)), torch.ones_like(lv), torch.sigmoid(lv))
        g, = torch.autograd.grad(ll.sum(), lv)
        assert np.allclose(ll.detach().numpy(), z[f'll_x{x}'], rtol=1e-6, atol=1e-7)
        assert np.allclose(g.numpy(), z[f'dll_dlogit_x{x}'], rtol=1e-5, atol=1e-7)
        nz = torch.from_numpy(z[f'dll_dlogit_x{x}']) != 0
        assert float(l[nz].min()) >= -15.9424 and float(l[nz].max()) <= 16.6356
        assert float(z[f'll_x{x}'].min()) > -15.9424
