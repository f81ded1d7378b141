#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference, build container only) on small seeded inputs.

Only the resulting vectors (inputs + expected outputs) are committed; the
reference source never enters this repo and this script is never run on the
GPU box.  Shims needed to import/run the reference here:
  * stub module ``nltk`` (imported, unused, not installed)
  * ``Distribution.set_default_validate_args(False)`` (x = -1 under mask 0)
  * PYTHONDONTWRITEBYTECODE so nothing is written under /root/reference

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import types

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = '/root/reference'


def import_reference():
    sys.modules.setdefault('nltk', types.SimpleNamespace(word_tokenize=None))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    torch.distributions.Distribution.set_default_validate_args(False)
    from src.torch_core import models as ref_models  # noqa
    from src import utils as ref_utils  # noqa
    return ref_models, ref_utils


def make_data(irt, B, I, A, missing, seed):
    """Synthetic 0/1 responses (-1 = missing) from a random IRT model."""
    g = torch.Generator().manual_seed(seed)
    D = {1: 1, 2: A + 1, 3: A + 2}[irt]
    theta = torch.randn(B, A, generator=g)
    item = torch.randn(I, D, generator=g)
    if irt == 1:
        logit = theta.sum(1, keepdim=True) + item.t()
    else:
        logit = theta @ (-item[:, :A].t()) + item[:, A:A + 1].t()
    p = torch.sigmoid(logit)
    if irt == 3:
        gs = torch.sigmoid(item[:, A + 1:A + 2]).t()
        p = gs + (1 - gs) * p
    resp = torch.bernoulli(p, generator=g)
    mask = torch.ones(B, I, dtype=torch.bool)
    if missing > 0:
        drop = torch.rand(B, I, generator=g) < missing
        # make sure at least one row is fully observed and one row is nearly empty
        drop[0] = False
        if B > 2:
            drop[1] = True
            drop[1, I // 2] = False
        resp = torch.where(drop, torch.full_like(resp, -1.0), resp)
        mask = ~drop
    return resp, mask


CASES = [
    # name, irt, A, B, I, cond, missing, drop, flows, beta, use_kl
    ('2pl_a1_uncond',            2, 1, 16, 20, False, 0.0, False, 0, 1.0, True),
    ('2pl_a1_config1_b16_i100',  2, 1, 16, 100, False, 0.0, False, 0, 1.0, True),      # BASELINE configs[0]: 100 items, A = 1, batch 16
    ('2pl_a1_uncond_beta05',     2, 1, 37, 95, False, 0.0, False, 0, 0.5, True),
    ('2pl_a8_uncond',            2, 8, 37, 130, False, 0.0, False, 0, 1.0, True),
    ('2pl_a2_uncond_miss_prior', 2, 2, 37, 95, False, 0.2, False, 0, 1.0, True),
    ('2pl_a1_uncond_miss_drop',  2, 1, 37, 95, False, 0.2, True, 0, 1.0, True),
    ('2pl_a8_uncond_miss_prior', 2, 8, 16, 130, False, 0.2, False, 0, 0.5, True),
    ('2pl_a3_uncond_nokl',       2, 3, 16, 20, False, 0.0, False, 0, 1.0, False),
    ('2pl_a1_uncond_miss_nokl',  2, 1, 16, 95, False, 0.2, False, 0, 1.0, False),
    ('1pl_a1_uncond',            1, 1, 16, 20, False, 0.0, False, 0, 1.0, True),
    ('1pl_a2_uncond_miss_prior', 1, 2, 37, 95, False, 0.2, False, 0, 1.0, True),
    ('3pl_a1_uncond',            3, 1, 37, 95, False, 0.0, False, 0, 1.0, True),
    ('3pl_a2_uncond_miss_drop',  3, 2, 16, 130, False, 0.2, True, 0, 0.5, True),
    ('3pl_a8_uncond_miss_prior', 3, 8, 16, 20, False, 0.2, False, 0, 1.0, True),
    ('2pl_a1_cond',              2, 1, 16, 20, True, 0.0, False, 0, 1.0, True),
    ('2pl_a2_cond_miss_prior',   2, 2, 37, 95, True, 0.2, False, 0, 1.0, True),
    ('3pl_a1_cond_miss_drop',    3, 1, 16, 130, True, 0.2, True, 0, 0.5, True),
    ('1pl_a1_cond',              1, 1, 16, 20, True, 0.0, False, 0, 1.0, True),
    ('2pl_a1_uncond_flows4',     2, 1, 16, 20, False, 0.0, False, 4, 1.0, False),
    ('2pl_a2_uncond_flows2_miss', 2, 2, 37, 95, False, 0.2, False, 2, 1.0, False),
    ('3pl_a1_cond_flows4',       3, 1, 16, 130, True, 0.0, False, 4, 1.0, False),
    ('3pl_a1_cond_flows4_miss',  3, 1, 37, 95, True, 0.2, False, 4, 1.0, False),
    ('2pl_a8_cond_miss_prior',   2, 8, 16, 20, True, 0.2, False, 0, 1.0, True),
    # --ability-dim above 8 (vibo.py:36-37 takes any int): the wave-per-person kernel's wide instantiation
    ('2pl_a12_uncond_miss_prior', 2, 12, 37, 130, False, 0.2, False, 0, 1.0, True),
    ('2pl_a9_cond_miss_prior',    2, 9, 16, 20, True, 0.2, False, 0, 1.0, True),
    ('2pl_a10_uncond_flows2',     2, 10, 16, 20, False, 0.0, False, 2, 1.0, False),
    ('1pl_a16_uncond_miss_drop',  1, 16, 16, 20, False, 0.2, True, 0, 0.5, True),
    # --ability-merge mean (models.py:584-594, 631-650): 12th field
    ('2pl_a1_uncond_mean',           2, 1, 16, 20, False, 0.0, False, 0, 1.0, True, 'mean'),
    ('2pl_a2_uncond_mean_miss',      2, 2, 37, 95, False, 0.2, False, 0, 0.5, True, 'mean'),
    ('3pl_a8_uncond_mean_miss',      3, 8, 16, 130, False, 0.2, False, 0, 1.0, True, 'mean'),
    ('1pl_a3_uncond_mean_flows2',    1, 3, 16, 20, False, 0.0, False, 2, 1.0, False, 'mean'),
    ('2pl_a1_uncond_mean_miss_nokl', 2, 1, 37, 95, False, 0.2, False, 0, 1.0, False, 'mean'),
    ('2pl_a2_cond_mean_miss',        2, 2, 37, 95, True, 0.2, False, 0, 1.0, True, 'mean'),
    ('3pl_a2_cond_mean_flows2',      3, 2, 16, 20, True, 0.0, False, 2, 1.0, False, 'mean'),
    # --generative-model link | deep | residual (models.py:769-919): 13th field
    ('2pl_a2_deep_miss',             2, 2, 16, 20, False, 0.2, False, 0, 1.0, True, 'product', 'deep'),
    ('3pl_a1_residual',              3, 1, 16, 70, False, 0.0, False, 0, 0.5, True, 'product', 'residual'),
    ('2pl_a1_link_miss',             2, 1, 37, 20, False, 0.2, False, 0, 1.0, True, 'product', 'link'),
    ('3pl_a2_link_flows2',           3, 2, 16, 20, False, 0.0, False, 2, 1.0, False, 'product', 'link'),
    ('1pl_a3_residual_mean_miss_drop', 1, 3, 16, 20, False, 0.2, True, 0, 1.0, True, 'mean', 'residual'),
    ('2pl_a8_deep_nokl',             2, 8, 16, 20, False, 0.0, False, 0, 1.0, False, 'product', 'deep'),
    ('2pl_a2_cond_residual_miss',    2, 2, 16, 20, True, 0.2, False, 0, 1.0, True, 'product', 'residual'),
]


def run_case(ref_models, case, out_dir):
    name, irt, A, B, I, cond, missing, drop, flows, beta, use_kl = case[:11]
    merge = case[11] if len(case) > 11 else 'product'
    gen = case[12] if len(case) > 12 else 'irt'
    seed = 1000 + sum(ord(c) for c in name)
    resp, mask = make_data(irt, B, I, A, missing, seed)
    cls = {1: ref_models.VIBO_1PL, 2: ref_models.VIBO_2PL, 3: ref_models.VIBO_3PL}[irt]
    torch.manual_seed(seed)
    model = cls(A, I, hidden_dim=64, ability_merge=merge, conditional_posterior=cond,
                generative_model=gen, response_dist='bernoulli',
                replace_missing_with_prior=not drop, n_norm_flows=flows)
    D = model.item_feat_dim

    # eps in the reference's draw order: item [I,D] first, then ability [B,A]
    torch.manual_seed(seed + 1)
    eps_item = torch.randn(I, D)
    eps_ability = torch.randn(B, A)
    torch.manual_seed(seed + 1)

    r3 = resp.unsqueeze(2)
    m3 = mask.long().unsqueeze(2)
    model.zero_grad()
    outs = model(r3, m3)
    if flows > 0:
        (_, _, response_mu, ability_k, ability, ability_mu, ability_logvar, a_ladj,
         item_k, item_feat, item_mu, item_lv, i_ladj) = outs
        loss = model.elbo(r3, m3, response_mu, ability, ability_mu, ability_logvar,
                          item_feat, item_mu, item_lv, annealing_factor=beta,
                          use_kl_divergence=False, ability_k=ability_k, item_feat_k=item_k,
                          ability_logabsdetjac=a_ladj, item_logabsdetjac=i_ladj)
    else:
        (_, _, response_mu, ability, ability_mu, ability_logvar,
         item_feat, item_mu, item_lv) = outs
        loss = model.elbo(*outs, annealing_factor=beta, use_kl_divergence=use_kl)
    loss.backward()

    # sanity: the eps replay really is what the reference drew
    chk = eps_ability * torch.exp(0.5 * ability_logvar) + ability_mu
    assert torch.allclose(chk, ability, atol=1e-6), name
    chk = eps_item * torch.exp(0.5 * item_lv) + item_mu
    assert torch.allclose(chk, item_feat, atol=1e-6), name

    rec = {
        'meta': json.dumps(dict(name=name, irt_model=irt, ability_dim=A, num_person=B, num_item=I,
                                conditional_posterior=cond, missing_frac=missing,
                                replace_missing_with_prior=not drop, n_norm_flows=flows,
                                annealing_factor=beta, use_kl_divergence=use_kl, ability_merge=merge, generative_model=gen,
                                hidden_dim=64, torch=torch.__version__)),
        'response': resp.numpy().astype(np.int8),
        'mask': mask.numpy().astype(np.uint8),
        'eps_item': eps_item.numpy(),
        'eps_ability': eps_ability.numpy(),
        'out.loss': loss.detach().numpy(),
        'out.ability': ability.detach().numpy(),
        'out.ability_mu': ability_mu.detach().numpy(),
        'out.ability_logvar': ability_logvar.detach().numpy(),
        'out.item_feat': item_feat.detach().numpy(),
        'out.response_mu': response_mu.detach().squeeze(2).numpy(),
    }
    if flows > 0:
        rec['out.ability_k'] = ability_k.detach().numpy()
        rec['out.ability_logabsdetjac'] = a_ladj.detach().numpy()
        rec['out.item_feat_k'] = item_k.detach().numpy()
        rec['out.item_feat_logabsdetjac'] = i_ladj.detach().numpy()
    for k, v in model.state_dict().items():
        rec['sd.' + k] = v.detach().numpy().copy()
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        rec['grad.' + k] = g.detach().numpy().copy()

    # parameters after 1 and 3 Adam steps (vibo.py:221, 243-268) re-using the same batch/eps
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    for step in range(3):
        torch.manual_seed(seed + 1)
        opt.zero_grad()
        outs = model(r3, m3)
        if flows > 0:
            (_, _, rmu, ak, a0, amu, alv, aladj, ik, i0, imu, ilv, iladj) = outs
            l = model.elbo(r3, m3, rmu, a0, amu, alv, i0, imu, ilv, annealing_factor=beta,
                           use_kl_divergence=False, ability_k=ak, item_feat_k=ik,
                           ability_logabsdetjac=aladj, item_logabsdetjac=iladj)
        else:
            l = model.elbo(*outs, annealing_factor=beta, use_kl_divergence=use_kl)
        l.backward()
        opt.step()
        if step in (0, 2):
            for k, v in model.state_dict().items():
                rec[f'adam{step + 1}.' + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(out_dir, f'case_{name}.npz'), **rec)
    return float(loss.detach())


VI_CASES = [
    # name, irt, A, B, P, I, missing, beta, use_kl   (un-amortized VI: models.py:100-243; persons index an embedding)
    ('vi_2pl_a2',          2, 2, 16, 40, 20, 0.0, 1.0, True),
    ('vi_3pl_a1_miss',     3, 1, 37, 60, 95, 0.2, 0.5, True),
    ('vi_1pl_a3_miss_nokl', 1, 3, 16, 16, 130, 0.2, 1.0, False),
]


def run_vi_case(ref_models, case, out_dir):
    name, irt, A, B, P, I, missing, beta, use_kl = case
    seed = 2000 + sum(ord(c) for c in name)
    resp, mask = make_data(irt, B, I, A, missing, seed)
    g = torch.Generator().manual_seed(seed)
    index = torch.randperm(P, generator=g)[:B]
    cls = {1: ref_models.VI_1PL, 2: ref_models.VI_2PL, 3: ref_models.VI_3PL}[irt]
    torch.manual_seed(seed)
    model = cls(A, P, I)
    with torch.no_grad():
        # N(0,1)-initialised per-person means AND log-variances put many cells into the Bernoulli probability-clamp band
        # (|logit| > ~12), where the reference's fp32 result is defined by its own rounding; move to the regime a fitted
        # model lives in (the parameters used are stored in the fixture, the outputs are still the reference's own)
        model.ability_mu_lookup.weight.mul_(0.5)
        model.ability_logvar_lookup.weight.mul_(0.3).sub_(2.0)
        model.item_mu_lookup.weight.mul_(0.6)
    D = model.item_feat_dim
    torch.manual_seed(seed + 1)
    eps_item = torch.randn(I, D)              # draw order of VI_1PL.encode: item first, then ability
    eps_ability = torch.randn(B, A)
    torch.manual_seed(seed + 1)
    r3, m3 = resp.unsqueeze(2), mask.long().unsqueeze(2)
    outs = model(index, r3, m3)
    (_, _, response_mu, ability, ability_mu, ability_logvar, item_feat, item_mu, item_lv) = outs
    loss = model.elbo(*outs, annealing_factor=beta, use_kl_divergence=use_kl)
    loss.backward()
    assert torch.allclose(eps_ability * torch.exp(0.5 * ability_logvar) + ability_mu, ability, atol=1e-6), name
    assert torch.allclose(eps_item * torch.exp(0.5 * item_lv) + item_mu, item_feat, atol=1e-6), name
    rec = {
        'meta': json.dumps(dict(name=name, irt_model=irt, ability_dim=A, num_person=P, batch=B, num_item=I, missing_frac=missing,
                                annealing_factor=beta, use_kl_divergence=use_kl, torch=torch.__version__)),
        'response': resp.numpy().astype(np.int8), 'mask': mask.numpy().astype(np.uint8), 'index': index.numpy(),
        'eps_item': eps_item.numpy(), 'eps_ability': eps_ability.numpy(),
        'out.loss': loss.detach().numpy(), 'out.ability': ability.detach().numpy(),
        'out.ability_mu': ability_mu.detach().numpy(), 'out.ability_logvar': ability_logvar.detach().numpy(),
        'out.item_feat': item_feat.detach().numpy(), 'out.response_mu': response_mu.detach().squeeze(2).numpy(),
    }
    for k, v in model.state_dict().items():
        rec['sd.' + k] = v.detach().numpy().copy()
    for k, p in model.named_parameters():
        rec['grad.' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
    np.savez_compressed(os.path.join(out_dir, f'{name}.npz'), **rec)
    return float(loss.detach())


MLE_CASES = [
    # name, irt, A, B, P, I, missing   (maximum-likelihood point estimates: models.py:22-97, loss of mle.py:192-197)
    ('mle_2pl_a2', 2, 2, 16, 40, 20, 0.0),
    ('mle_3pl_a1_miss', 3, 1, 37, 60, 95, 0.2),
    ('mle_1pl_a3_miss', 1, 3, 16, 16, 130, 0.2),
]


def run_mle_case(ref_models, case, out_dir):
    import torch.nn.functional as F
    name, irt, A, B, P, I, missing = case
    seed = 3000 + sum(ord(c) for c in name)
    resp, mask = make_data(irt, B, I, A, missing, seed)
    g = torch.Generator().manual_seed(seed)
    index = torch.randperm(P, generator=g)[:B]
    cls = {1: ref_models.MLE_1PL, 2: ref_models.MLE_2PL, 3: ref_models.MLE_3PL}[irt]
    torch.manual_seed(seed)
    model = cls(A, P, I)
    with torch.no_grad():                       # a regime without saturated logits (see run_vi_case)
        model.ability.weight.mul_(0.5)
        model.item_feat.weight.mul_(0.6)
    r3, m3 = resp.unsqueeze(2), mask.long().unsqueeze(2)
    response_mu = model(index, r3, m3)
    # mle.py:194-196 with the missing cells' target (-1 in the loaders) set to 0: those terms are multiplied by mask = 0
    loss = (F.binary_cross_entropy(response_mu, r3.clamp(min=0).float(), reduction='none') * m3).mean()
    loss.backward()
    rec = {
        'meta': json.dumps(dict(name=name, irt_model=irt, ability_dim=A, num_person=P, batch=B, num_item=I, missing_frac=missing,
                                torch=torch.__version__)),
        'response': resp.numpy().astype(np.int8), 'mask': mask.numpy().astype(np.uint8), 'index': index.numpy(),
        'out.loss': loss.detach().numpy(), 'out.response_mu': response_mu.detach().squeeze(2).numpy(),
    }
    for k, v in model.state_dict().items():
        rec['sd.' + k] = v.detach().numpy().copy()
    for k, p in model.named_parameters():
        rec['grad.' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
    np.savez_compressed(os.path.join(out_dir, f'{name}.npz'), **rec)
    return float(loss.detach())


def saturation_case(ref_utils, out_dir):
    """masked_bernoulli_log_pdf(sigmoid(l)) and its gradient for l in [-30, 30]
    plus dense sweeps around the clamp thresholds (utils.py:46-49)."""
    l = torch.cat([
        torch.linspace(-30, 30, 2401, dtype=torch.float64),
        torch.linspace(-15.96, -15.92, 801, dtype=torch.float64),
        torch.linspace(16.60, 16.66, 1201, dtype=torch.float64),
    ]).float()
    rec = {'logit': l.numpy()}
    for x in (0.0, 1.0):
        lv = l.clone().requires_grad_(True)
        ll = ref_utils.masked_bernoulli_log_pdf(torch.full_like(lv, x), torch.ones_like(lv),
                                                torch.sigmoid(lv))
        g, = torch.autograd.grad(ll.sum(), lv)
        rec[f'll_x{int(x)}'] = ll.detach().numpy()
        rec[f'dll_dlogit_x{int(x)}'] = g.numpy()
    np.savez_compressed(os.path.join(out_dir, 'saturation.npz'), **rec)


def saturation_3pl_case(ref_models, ref_utils, out_dir):
    """3PL inside and around the Bernoulli probability clamp: p = g + (1 - g) sigmoid(l) through the reference's own
    irt_model_3pl (models.py:748-766) and masked_bernoulli_log_pdf (utils.py:46-49), one person with theta = 0 so that the
    logit of item i is its difficulty.  Stored per guess logit and response value: p (fp32, as the reference rounds it),
    the log-likelihood and its gradients w.r.t. the difficulty and the guess logit."""
    l = torch.cat([torch.linspace(-30, 30, 1201, dtype=torch.float64),
                   torch.linspace(13.0, 19.0, 1201, dtype=torch.float64)]).float()
    rec = {'logit': l.numpy(), 'guess_logit': np.asarray([-4.0, 0.0, 3.0], dtype=np.float32)}
    for gi, gl in enumerate(rec['guess_logit']):
        for x in (0.0, 1.0):
            item = torch.stack([torch.ones_like(l), l, torch.full_like(l, float(gl))], dim=1).requires_grad_(True)
            ability = torch.zeros(1, 1)
            p = ref_models.irt_model_3pl(ability, item)                      # [1, I, 1]
            ll = ref_utils.masked_bernoulli_log_pdf(torch.full_like(p, x), torch.ones_like(p), p)
            g, = torch.autograd.grad(ll.sum(), item)
            rec[f'p_g{gi}'] = p.detach().reshape(-1).numpy()
            rec[f'll_g{gi}_x{int(x)}'] = ll.detach().reshape(-1).numpy()
            rec[f'dll_db_g{gi}_x{int(x)}'] = g[:, 1].numpy()
            rec[f'dll_dguess_g{gi}_x{int(x)}'] = g[:, 2].numpy()
    np.savez_compressed(os.path.join(out_dir, 'saturation_3pl.npz'), **rec)


def seeded_init_case(ref_models, out_dir):
    """state_dict of the reference classes right after construction under torch.manual_seed: constructor order and
    weights_init (models.py:281-329, 512-518) -- the drop-in classes must draw the same numbers."""
    rec = {}
    cfgs = [('1pl_a1', 1, 1, 12, 'product', False, 0), ('2pl_a2_cond', 2, 2, 12, 'product', True, 0),
            ('3pl_a1_flows2', 3, 1, 12, 'product', False, 2), ('2pl_a3_mean', 2, 3, 12, 'mean', False, 0)]
    rec['meta'] = json.dumps([dict(name=n, irt_model=i, ability_dim=a, num_item=I, ability_merge=m, conditional_posterior=c,
                                   n_norm_flows=f, seed=100 + k) for k, (n, i, a, I, m, c, f) in enumerate(cfgs)])
    for k, (n, irt, A, I, merge, cond, flows) in enumerate(cfgs):
        torch.manual_seed(100 + k)
        cls = {1: ref_models.VIBO_1PL, 2: ref_models.VIBO_2PL, 3: ref_models.VIBO_3PL}[irt]
        model = cls(A, I, ability_merge=merge, conditional_posterior=cond, n_norm_flows=flows)
        for key, v in model.state_dict().items():
            rec[f'{n}.{key}'] = v.detach().numpy().copy()
        rec[f'{n}.next_randn'] = torch.randn(4).numpy()            # the generator state after construction
    np.savez_compressed(os.path.join(out_dir, 'seeded_init.npz'), **rec)


def critlangacq_case(out_dir):
    """The reference's CritLangAcq loader (datasets.py:283-440) on a synthetic data.csv: 95 q* columns in the file's own
    (scrambled) order plus metadata columns, 503 rows, some cells -1.  Stored: the csv's columns as arrays (the test rebuilds
    the file), and what the reference loads for train / test, with and without max_num_person / max_num_item."""
    import tempfile
    import pandas as pd
    from src import datasets as ref_ds
    rs = np.random.RandomState(11)
    keys = ['q1', 'q2', 'q3', 'q5', 'q6', 'q7', 'q9_1', 'q9_4', 'q10_2', 'q10_4', 'q11_3', 'q11_4', 'q12_1', 'q12_2', 'q12_4',
            'q13_3', 'q13_4', 'q14_3', 'q14_4', 'q15_1', 'q15_2', 'q15_3', 'q16_3', 'q16_4', 'q17_1', 'q17_3', 'q17_4', 'q18_2',
            'q18_3', 'q18_4', 'q19_1', 'q19_2', 'q19_3', 'q19_4', 'q20_1', 'q20_2', 'q20_3', 'q20_4', 'q21_1', 'q21_2', 'q21_3',
            'q21_4', 'q22_1', 'q22_2', 'q22_3', 'q22_4', 'q23_3', 'q23_4', 'q24_1', 'q24_2', 'q24_3', 'q24_4', 'q25_1', 'q25_2',
            'q25_3', 'q25_4', 'q26_1', 'q26_2', 'q26_3', 'q26_4', 'q27_1', 'q27_2', 'q27_3', 'q27_4', 'q28_1', 'q28_2', 'q29_1',
            'q29_2', 'q29_3', 'q29_4', 'q30_1', 'q30_2', 'q30_3', 'q30_4', 'q31_1', 'q31_4', 'q32_5', 'q32_6', 'q32_8', 'q33_4',
            'q33_5', 'q33_6', 'q33_7', 'q34_1', 'q34_2', 'q34_3', 'q34_4', 'q34_6', 'q34_8', 'q35_1', 'q35_2', 'q35_4', 'q35_5',
            'q35_7', 'q35_8']
    n = 503
    cols = {'id': np.arange(n), 'age': rs.randint(7, 80, n), 'education': rs.randint(0, 6, n)}
    file_order = list(keys)
    rs.shuffle(file_order)                              # the loader must select by name, not by position
    file_order.insert(7, 'q4_decoy')                    # a q-column that is NOT one of the 95 items
    for k in file_order:
        v = (rs.rand(n) < 0.6).astype(np.int64)
        v[rs.rand(n) < 0.03] = -1
        cols[k] = v
    df = pd.DataFrame(cols)
    rec = {'columns': json.dumps(list(df.columns)), 'table': df.to_numpy().astype(np.int64)}
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'critlangacq'))
        df.to_csv(os.path.join(tmp, 'critlangacq', 'data.csv'), index=False)
        ref_ds.CHILDREN_LANG_DIR = os.path.join(tmp, 'critlangacq')
        for name, kw in (('train', dict(train=True)), ('test', dict(train=False)),
                         ('train_cap', dict(train=True, max_num_person=100, max_num_item=40))):
            d = ref_ds.Children_LanguageAcquisition(**kw)
            rec[f'{name}.response'] = np.asarray(d.response).astype(np.int64)
            rec[f'{name}.mask'] = np.asarray(d.mask).astype(np.int64)
            rec[f'{name}.item_id'] = np.asarray(d.item_id).astype(np.int64)
            idx, r, iid, m = d[3]
            rec[f'{name}.getitem3.response'] = r.numpy()
            rec[f'{name}.getitem3.item_id'] = iid.numpy()
            rec[f'{name}.getitem3.mask'] = m.numpy()
    np.savez_compressed(os.path.join(out_dir, 'critlangacq_loader.npz'), **rec)


def log_marginal_case(ref_models, out_dir):
    """log_marginal (models.py:445-504) with S=8 under a fixed seed; the eps
    sequence (item then ability, per sample) is stored so it can be replayed."""
    for name, irt, A, B, I, cond, missing, flows in (
            ('logmarg_2pl_a2', 2, 2, 16, 20, False, 0.2, 0),
            ('logmarg_3pl_a1_cond_flows2', 3, 1, 16, 20, True, 0.0, 2)):
        seed = 77 + len(name)
        resp, mask = make_data(irt, B, I, A, missing, seed)
        cls = {1: ref_models.VIBO_1PL, 2: ref_models.VIBO_2PL, 3: ref_models.VIBO_3PL}[irt]
        torch.manual_seed(seed)
        model = cls(A, I, ability_merge='product', conditional_posterior=cond, n_norm_flows=flows)
        D = model.item_feat_dim
        S = 8
        torch.manual_seed(seed + 5)
        ei, ea = [], []
        for _ in range(S):
            ei.append(torch.randn(I, D))
            ea.append(torch.randn(B, A))
        torch.manual_seed(seed + 5)
        logp = model.log_marginal(resp.unsqueeze(2), mask.long().unsqueeze(2), num_samples=S)
        rec = {
            'meta': json.dumps(dict(name=name, irt_model=irt, ability_dim=A, num_person=B, num_item=I,
                                    conditional_posterior=cond, missing_frac=missing,
                                    replace_missing_with_prior=True, n_norm_flows=flows,
                                    num_samples=S, hidden_dim=64)),
            'response': resp.numpy().astype(np.int8), 'mask': mask.numpy().astype(np.uint8),
            'eps_item': torch.stack(ei).numpy(), 'eps_ability': torch.stack(ea).numpy(),
            'out.logp': logp.detach().numpy(),
        }
        for k, v in model.state_dict().items():
            rec['sd.' + k] = v.detach().numpy()
        np.savez_compressed(os.path.join(out_dir, f'{name}.npz'), **rec)


def artificial_mask_case(out_dir):
    """artificially_mask_dataset of the reference (datasets.py:46-78) on a 200 x 95 matrix, perc 0.2."""
    from src.datasets import artificially_mask_dataset

    class D:
        pass
    rs = np.random.RandomState(7)
    d = D()
    d.response = (rs.rand(200, 95) < 0.6).astype(np.int64)
    d.mask = np.ones_like(d.response)
    out = artificially_mask_dataset(d, 0.2)
    np.savez_compressed(os.path.join(out_dir, 'artificial_mask.npz'), response=d.response, mask=d.mask,
                        missing_indices=out.missing_indices, missing_labels=out.missing_labels,
                        masked_mask=out.mask, masked_response=out.response)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--only', default='', help='only (re)generate the ELBO cases whose name contains this substring')
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)
    ref_models, ref_utils = import_reference()
    for case in CASES:
        if args.only and args.only not in case[0]:
            continue
        loss = run_case(ref_models, case, out_dir)
        print(f'{case[0]:34s} loss={loss:.6f}')
    for case in VI_CASES:
        if args.only and args.only not in case[0]:
            continue
        print(f'{case[0]:34s} loss={run_vi_case(ref_models, case, out_dir):.6f}')
    for case in MLE_CASES:
        if args.only and args.only not in case[0]:
            continue
        print(f'{case[0]:34s} loss={run_mle_case(ref_models, case, out_dir):.6f}')
    if not args.only:
        saturation_case(ref_utils, out_dir)
        saturation_3pl_case(ref_models, ref_utils, out_dir)
        seeded_init_case(ref_models, out_dir)
        critlangacq_case(out_dir)
        log_marginal_case(ref_models, out_dir)
        artificial_mask_case(out_dir)
    print('wrote', out_dir)


if __name__ == '__main__':
    main()
