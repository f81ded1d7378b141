"""Host logic of the drop-in modules on CPU: state_dict compatibility with the
reference, loss / gradient composition around the fused op, the reference call
pattern ``model.elbo(*model(response, mask))`` -- with the native entry points
replaced by the CPU analytic restatement (oracle/cpu_backend.py).  The same
assertions run against the real HIP kernel in tests/test_gpu_parity.py."""
import pytest
import torch

import os

from conftest import GOLDEN_DIR, Golden, rel_err
from oracle import cpu_backend
from oracle import vibo_oracle as O
from oracle import vibo_table_ref as T
from vibo_amd import ops
from vibo_amd.torch_core.models import VIBO_1PL, VIBO_2PL, VIBO_3PL, DeferredResponseMu

CLS = {1: VIBO_1PL, 2: VIBO_2PL, 3: VIBO_3PL}


@pytest.fixture()
def cpu_ops():
    restore = cpu_backend.install(ops)
    yield
    restore()


def build_model(golden):
    m = golden.meta
    model = CLS[m['irt_model']](m['ability_dim'], m['num_item'], hidden_dim=m['hidden_dim'],
                                ability_merge=m.get('ability_merge', 'product'), conditional_posterior=m['conditional_posterior'],
                                replace_missing_with_prior=m['replace_missing_with_prior'],
                                n_norm_flows=m['n_norm_flows'], generative_model=m.get('generative_model', 'irt'))
    model.load_state_dict(golden.sd, strict=True)      # same keys and shapes as the reference
    return model


def run_reference_pattern(model, golden, mask_dtype=torch.int64):
    """vibo.py:239-267 call pattern with the golden's eps replayed."""
    m = golden.meta
    response = golden.response.unsqueeze(2)
    mask = golden.mask.to(mask_dtype).unsqueeze(2)
    outs = model(response, mask, eps_item=golden.eps_item, eps_ability=golden.eps_ability)
    if m['n_norm_flows'] > 0:
        (r, k, rmu, ak, a0, amu, alv, aladj, ik, i0, imu, ilv, iladj) = outs
        loss = model.elbo(r, k, rmu, a0, amu, alv, i0, imu, ilv, annealing_factor=m['annealing_factor'],
                          use_kl_divergence=False, ability_k=ak, item_feat_k=ik,
                          ability_logabsdetjac=aladj, item_logabsdetjac=iladj)
    else:
        loss = model.elbo(*outs, annealing_factor=m['annealing_factor'],
                          use_kl_divergence=m['use_kl_divergence'])
    return outs, loss


def check_against_golden(model, golden, outs, loss, tol_loss=1e-4, tol_grad=3e-4, tol_truth=1.5e-3, strict=False):
    m = golden.meta
    assert rel_err(loss.detach(), golden.out['loss']) < tol_loss
    flows = m['n_norm_flows'] > 0
    amu, alv, a0 = (outs[5], outs[6], outs[4]) if flows else (outs[4], outs[5], outs[3])
    assert (amu.cpu() - golden.out['ability_mu']).abs().max() < 2e-5 * max(1.0, float(golden.out['ability_mu'].abs().max()))
    assert (alv.cpu() - golden.out['ability_logvar']).abs().max() < 2e-5 * max(1.0, float(golden.out['ability_logvar'].abs().max()))
    assert (a0.cpu() - golden.out['ability']).abs().max() < 5e-5 * max(1.0, float(golden.out['ability'].abs().max()))
    if flows:
        assert (outs[3].cpu() - golden.out['ability_k']).abs().max() < 1e-4
        assert (outs[7].cpu() - golden.out['ability_logabsdetjac']).abs().max() < 1e-4
    loss.backward()
    # The reference computes in fp32 and its own gradients carry rounding noise (up to ~8e-4 of the
    # tensor's max on 3PL cases: the probability clamp + log).  Allow  tol + |golden - fp64 oracle|.
    sd64 = {k: v.double() for k, v in golden.sd.items()}
    _, truth = O.elbo_loss_and_grads(sd64, golden.response.cpu().double(), golden.mask.cpu(),
                                     golden.eps_item.cpu().double(), golden.eps_ability.cpu().double(),
                                     **golden.cfg)
    for name, p in model.named_parameters():
        g_ref = golden.grad[name]
        g = p.grad.cpu() if p.grad is not None else torch.zeros_like(g_ref)
        scale = float(g_ref.abs().max())
        if scale == 0.0:
            assert float(g.abs().max()) < 1e-6, name
        else:
            # the fp32 CPU stand-in is as noisy as the reference (whose own fp32 gradients sit up to 1.6e-2 from fp64 on
            # the mean-merge encoder's first layer): allow the reference's own distance from the fp64 oracle on top
            if strict:
                # the HIP path (tests/test_gpu_parity.py): within tol of the exact (fp64) gradient, or -- where the reference's
                # own fp32 arithmetic sits further from it than that (3PL cells inside the probability clamp band) -- within
                # tol of the reference's gradient.  No allowance added on top of either.
                e_truth, e_ref = rel_err(g, truth[name]), rel_err(g, g_ref)
                ref_off = rel_err(g_ref, truth[name])
                # (one golden, 3pl_a8_uncond_mean_miss: the reference's own fp32 gradients are 3-18 % away from the exact ones --
                #  saturated 3PL cells; a tensor the reference gets that wrong is held to 6 % of the reference's own error instead)
                # Measured on the GPU over all goldens x kernel pins (gpurun_out/r6_tolerances_golden.jsonl, round 6): every tensor of
                # every golden but one is within 2.9e-5; the one is 3pl_a8_uncond_mean_miss, whose cells sit inside the probability
                # clamp band -- there the measured distance to the reference is <= 0.053 x the reference's own distance to fp64.
                # (2pl_a10_uncond_flows2 has such cells too -- flows push the sample out: its reference gradients are up to 4 % from
                #  fp64 -- but the kernel stays within 2.9e-5 of the reference there)
                clamp_band = ref_off >= 1e-2
                tol = tol_grad if not clamp_band else max(tol_grad, 0.06 * ref_off)
                if os.environ.get('VIBO_TOL_RECORD'):
                    import json
                    with open(os.environ['VIBO_TOL_RECORD'], 'a') as f:
                        f.write(json.dumps({'kind': 'golden:' + name, 'err': min(e_truth, e_ref), 'e_truth': e_truth, 'e_ref': e_ref, 'ref_off': ref_off,
                                            'tol': tol, 'irt': m['irt_model'], 'test': os.environ.get('PYTEST_CURRENT_TEST', '')}) + '\n')
                assert min(e_truth, e_ref) < tol, (name, e_truth, e_ref, ref_off)
                continue
            assert rel_err(g, truth[name]) < tol_truth + rel_err(g_ref, truth[name]), name
            assert rel_err(g, g_ref) < tol_grad + rel_err(g_ref, truth[name]), name


def test_state_dict_keys_match_reference(golden):
    model = build_model(golden)
    assert list(model.state_dict().keys()) == list(golden.sd.keys())


def test_reference_call_pattern_matches_golden(cpu_ops, golden):
    model = build_model(golden)
    outs, loss = run_reference_pattern(model, golden)
    assert isinstance(outs[2], DeferredResponseMu)
    check_against_golden(model, golden, outs, loss)


def test_mask_dtypes_equivalent(cpu_ops, golden):
    model = build_model(golden)
    _, l_i64 = run_reference_pattern(model, golden, torch.int64)
    _, l_bool = run_reference_pattern(model, golden, torch.bool)
    _, l_u8 = run_reference_pattern(model, golden, torch.uint8)
    assert float(l_i64.detach()) == float(l_bool.detach()) == float(l_u8.detach())


def test_cell_codes_rows_match_golden(cpu_ops, golden):
    """Format P host plumbing (ops.CellCodes -> prepare_rows -> VIBO_MASK_CODES): the goldens fed as one byte per cell."""
    model = build_model(golden)
    mask = golden.mask != 0
    I = golden.response.shape[1]
    padded = torch.full((golden.response.shape[0], (I + 3) // 4 * 4), 2, dtype=torch.uint8)
    padded[:, :I] = torch.where(mask, (golden.response == 1).to(torch.uint8), torch.full_like(mask, 2, dtype=torch.uint8))
    codes = ops.CellCodes(padded[:, :I])
    r2, m2 = codes.unpack()
    assert torch.equal(m2, mask) and torch.equal(r2[m2], golden.response[m2])
    assert torch.equal(codes.rows(torch.tensor([1, 0])).codes, codes.codes[[1, 0]])
    with pytest.raises(ValueError):
        ops.prepare_rows(codes, mask)
    with pytest.raises(ValueError):
        ops.CellCodes(padded[:, :I].float())
    m = golden.meta
    outs = model(codes, None, eps_item=golden.eps_item, eps_ability=golden.eps_ability)
    if m['n_norm_flows'] > 0:
        (r, k, rmu, ak, a0, amu, alv, aladj, ik, i0, imu, ilv, iladj) = outs
        loss = model.elbo(r, k, rmu, a0, amu, alv, i0, imu, ilv, annealing_factor=m['annealing_factor'],
                          use_kl_divergence=False, ability_k=ak, item_feat_k=ik,
                          ability_logabsdetjac=aladj, item_logabsdetjac=iladj)
    else:
        loss = model.elbo(*outs, annealing_factor=m['annealing_factor'], use_kl_divergence=m['use_kl_divergence'])
    check_against_golden(model, golden, outs, loss)


def test_decode_and_deferred_response_mu(cpu_ops, golden):
    model = build_model(golden)
    outs, _ = run_reference_pattern(model, golden)
    rmu = outs[2].materialize()
    assert rmu.shape == (golden.meta['num_person'], golden.meta['num_item'], 1)
    assert (rmu.squeeze(2) - golden.out['response_mu']).abs().max() < 1e-5


def test_table_ref_matches_autograd_oracle(golden):
    """The analytic backward the kernel implements == autograd through the naive oracle."""
    cfg = golden.cfg
    if cfg['n_norm_flows'] > 0:
        mode = 'sampled'
    else:
        mode = 'kl' if cfg['use_kl_divergence'] else 'sampled'
    sd = {k: v.double() for k, v in golden.sd.items()}
    A = cfg['ability_dim']
    item_mu, item_lv = sd['item_encoder.mu_lookup.weight'], sd['item_encoder.logvar_lookup.weight']
    item_feat = golden.eps_item.double() * torch.exp(0.5 * item_lv) + item_mu
    flows = None
    item_k = item_feat
    if cfg['n_norm_flows'] > 0:
        flows = [(T.flow_uhat(sd[f'ability_norm_flows.flows.{k}.u'], sd[f'ability_norm_flows.flows.{k}.w']),
                  sd[f'ability_norm_flows.flows.{k}.w'], sd[f'ability_norm_flows.flows.{k}.b'])
                 for k in range(cfg['n_norm_flows'])]
        item_k, _ = O.planar_flows(sd, 'item_norm_flows', item_feat, cfg['n_norm_flows'])
    mean = golden.meta.get('ability_merge', 'product') == 'mean'
    if mean:      # the kernel takes the per-person posterior as given (VIBO_POSTERIOR_GIVEN): table = [B, 2A]
        table = torch.cat(O.ability_posterior(sd, golden.response.double(), golden.mask, item_feat, ability_dim=A,
                                              conditional_posterior=cfg['conditional_posterior'],
                                              replace_missing_with_prior=True), dim=1)
        table = table.detach().requires_grad_(True)
    else:
        table = T.encoder_table(sd, item_feat, cfg['conditional_posterior']).detach().requires_grad_(True)
    item_leaf = item_k.detach().requires_grad_(True)
    out = T.fused_elbo_ref(table.detach(), item_leaf.detach(), golden.response.double(), golden.mask,
                           golden.eps_ability.double(), irt_model=cfg['irt_model'], ability_dim=A,
                           conditional_posterior=cfg['conditional_posterior'] and not mean,
                           replace_missing_with_prior=cfg['replace_missing_with_prior'], mode=mode,
                           flow_uhat_w_b=flows, exact_saturation=False, given_posterior=mean)

    # autograd version of the same heads from the op-by-op oracle pieces
    B, I = golden.response.shape
    resp, mask = golden.response.double(), golden.mask
    x = (resp == 1).long()
    k = (mask != 0).double().unsqueeze(2)
    if mean:
        m = s = None
    elif cfg['conditional_posterior']:
        idx = torch.arange(I).unsqueeze(0).expand(B, I)
        m, s = table[x, idx, :A], table[x, idx, A:]
    else:
        m, s = table[x][..., :A], table[x][..., A:]
    if mean:
        amu, alv = table[:, :A], table[:, A:]
    elif cfg['replace_missing_with_prior']:
        amu, alv = O.product_of_experts((m * k).permute(1, 0, 2), (s * k).permute(1, 0, 2))
    else:
        amu, alv = O.product_of_experts(m.permute(1, 0, 2), s.permute(1, 0, 2), weight=k.permute(1, 0, 2))
    theta0 = golden.eps_ability.double() * torch.exp(0.5 * alv) + amu
    theta, ladj = theta0, torch.zeros(B, dtype=torch.float64)
    if flows:
        for (uhat, w, b) in flows:
            t = torch.tanh(theta @ w + b)
            ladj = ladj + torch.log(torch.abs(1 + (1 - t * t) * torch.dot(w, uhat)) + 1e-8)
            theta = theta + uhat.unsqueeze(0) * t.unsqueeze(1)
    probs = O.irt_link(cfg['irt_model'], theta, item_leaf)
    ll = O.masked_bernoulli_ll(resp, mask, probs).sum()
    if mode == 'kl':
        reg = O.kl_std_normal(amu, alv).sum()
    else:
        reg = O.normal_logpdf(theta0, amu, alv).sum() - ladj.sum() - O.std_normal_logpdf(theta).sum()
    if mean and cfg['irt_model'] == 3 and float(out['logit'].abs().max()) > 12.0:
        # 3PL clamps p itself at fp32 eps (utils.py:46-49 through torch's Bernoulli): inside that band the fp64 autograd
        # oracle (clamp at fp64 eps) is not the same function; the saturation golden covers the band
        pytest.skip('saturated 3PL logits: analytic-vs-autograd identity does not apply')
    assert rel_err(out['ll'], ll.detach()) < 1e-9
    assert rel_err(out['reg'], reg.detach()) < 1e-9
    g_t0, g_i0 = torch.autograd.grad(ll, [table, item_leaf], retain_graph=True)
    g_t1, = torch.autograd.grad(reg, [table], retain_graph=True)
    assert rel_err(out['g_table'][0], g_t0) < 1e-8
    assert rel_err(out['g_table'][1], g_t1) < 1e-8
    assert rel_err(out['g_item'], g_i0) < 1e-8


@pytest.mark.parametrize('name', ['logmarg_2pl_a2', 'logmarg_3pl_a1_cond_flows2'])
def test_log_marginal_matches_reference_golden(name, cpu_ops):
    """model.log_marginal (models.py:445-504) under the reference's recorded noise sequence; host logic on the CPU
    stand-in backend (one forward per sample)."""
    g = Golden(os.path.join(GOLDEN_DIR, name + '.npz'))
    model = build_model(g)
    logp = model.log_marginal(g.response.unsqueeze(2), g.mask.long().unsqueeze(2), num_samples=g.meta['num_samples'],
                              eps_item=g.eps_item, eps_ability=g.eps_ability)
    ref = float(g.out['logp'])
    assert abs(float(logp) - ref) < 1e-4 * max(1.0, abs(ref))


def test_elbo_accepts_a_materialised_response_mu_and_reruns_with_the_row_index(cpu_ops=None):
    """(a) a caller that builds response_mu through decode() before elbo() (the reference's predictive code does) gets the
    fused step's loss; (b) elbo(use_kl_divergence=False) after a KL-mode forward on gathered rows re-runs the step on the
    same rows with the same item noise."""
    from oracle import cpu_backend
    from vibo_amd import ops
    from vibo_amd.torch_core.models import VIBO_2PL
    restore = cpu_backend.install(ops)
    try:
        torch.manual_seed(3)
        B, I, A = 40, 24, 2
        model = VIBO_2PL(A, I, ability_merge='product')
        resp = (torch.rand(B, I) < 0.5).float()
        mask = torch.rand(B, I) > 0.2
        rows = torch.tensor([5, 1, 17, 30, 2, 9, 11, 39])
        outs = model(resp, mask, row_index=rows)
        loss_a = model.elbo(*outs)
        rmu = model.decode(outs[3], outs[6])
        loss_b = model.elbo(outs[0], outs[1], rmu, *outs[3:])
        assert torch.equal(loss_a, loss_b)
        # KL-mode forward, sampled-mode elbo with gradients on: the step is redone on the SAME rows and item noise
        loss_c = model.elbo(*outs, use_kl_divergence=False)
        eps_i = model._last_ctx.eps_item
        outs2 = model(resp[rows], mask[rows], eps_item=eps_i, eps_ability=model._last_ctx.eps_ability)
        loss_d = model.elbo(*outs2, use_kl_divergence=False)
        assert abs(float(loss_c) - float(loss_d)) <= 1e-4 * abs(float(loss_d))
    finally:
        restore()
