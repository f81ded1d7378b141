"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the VIBO amortized-ELBO hot path.

This file is a from-scratch CPU restatement (PyTorch-CPU, autograd for the
gradients) of the reference algorithm, following the reference's *op sequence*
(per-(person,item) encoder MLP -> product of experts -> reparameterised sample
-> 1PL/2PL/3PL link -> masked Bernoulli log-lik + KL).  It is the checker for
the HIP path and the timed "port" CPU baseline of bench.py.  Nothing in the
product package may import it: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg do.

Parity pinning: the reference ships no tests / golden vectors,
so this oracle is pinned against outputs of the reference itself, generated in
the build container by tools/gen_golden.py (which imports /root/reference) and
committed as tests/golden/*.npz.  tests/test_oracle_golden.py checks every
fixture.

Reference lines each function follows (paths relative to the reference repo):
  encoder_mlp            src/torch_core/models.py:575-582, 599
  product_of_experts     src/utils.py:105-113
  ability_posterior      src/torch_core/models.py:596-629 (uncond), 695-710 (cond)
  irt_link               src/torch_core/models.py:729-766
  decoder_probs          src/torch_core/models.py:769-919 (LinkedIRT / DeepIRT / ResidualIRT)
  planar_flows           src/torch_core/flows.py:21-41, 58-66
  masked_bernoulli_ll    src/utils.py:46-49 (+ torch.distributions.Bernoulli
                         probs->logits clamp, binary_cross_entropy_with_logits)
  kl_std_normal          src/utils.py:85-88
  normal_logpdf          src/utils.py:59-67
  elbo                   src/torch_core/models.py:380-443
  forward                src/torch_core/models.py:337-371, 506-510

Parameters are passed as a dict keyed by the reference's ``state_dict`` names
(``ability_encoder.mlp.0.weight`` ... ``item_encoder.mu_lookup.weight`` ...
``ability_norm_flows.flows.0.u`` ...), so a golden fixture's state_dict feeds
straight in.
"""
import math

import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


def item_feat_dim(irt_model, ability_dim):
    """models.py:331-332, 523-524, 538-539."""
    return {1: 1, 2: ability_dim + 1, 3: ability_dim + 2}[int(irt_model)]


def encoder_mlp(params, x, prefix='ability_encoder.mlp'):
    """Linear -> ELU -> Linear -> ELU -> Linear on rows of x."""
    # nn.ELU(inplace=True) in the reference (models.py:575-582): the same in-place kernels forward and backward
    h = F.elu(F.linear(x, params[f'{prefix}.0.weight'], params[f'{prefix}.0.bias']), inplace=True)
    h = F.elu(F.linear(h, params[f'{prefix}.2.weight'], params[f'{prefix}.2.bias']), inplace=True)
    return F.linear(h, params[f'{prefix}.4.weight'], params[f'{prefix}.4.bias'])


def product_of_experts(mu, logvar, weight=None, eps=1e-8):
    """Experts along dim 0.  ``weight`` (0/1) drops experts (the --drop-missing
    path keeps only observed experts; summing with a 0 weight is the same as
    boolean-indexing them away)."""
    prec = 1.0 / (torch.exp(logvar) + eps)
    if weight is not None:
        prec = prec * weight
    total = prec.sum(0)
    return (mu * prec).sum(0) / total, torch.log(1.0 / total)


def ability_posterior(params, response, mask, item_feat, *, ability_dim,
                      conditional_posterior, replace_missing_with_prior):
    """q(ability | responses[, items]) for every person: [B,A] mu, logvar.

    response [B,I] float (0/1, -1 = missing), mask [B,I] (1 = observed).
    The reference replaces the experts of missing cells by N(0,1) prior experts
    (mu=0, logvar=0) or drops them; both are order-independent sums, so the
    per-person python loop of models.py:606-625 reduces to the masked tensor
    expression below.
    """
    B, I = response.shape
    x = response.reshape(B * I, 1)
    if conditional_posterior:
        feat = item_feat.unsqueeze(0).expand(B, I, item_feat.shape[1]).reshape(B * I, -1)
        x = torch.cat([x, feat], dim=1)
    if 'ability_encoder.mlp1.0.weight' in params:
        # --ability-merge mean (models.py:584-594, 631-650): per-term features elu(mlp1(x)), mean over the OBSERVED
        # items of the person (plain mean when nothing is missing), then mlp2 on the [B, H] means.  The
        # replace-with-prior switch plays no role here.
        pre = 'ability_encoder.mlp1'
        h = F.elu(F.linear(x, params[f'{pre}.0.weight'], params[f'{pre}.0.bias']))
        hid = F.elu(F.linear(h, params[f'{pre}.2.weight'], params[f'{pre}.2.bias'])).reshape(B, I, -1)
        obs = mask.to(hid.dtype).unsqueeze(2)
        hid_mean = (hid * obs).sum(1) / obs.sum(1)
        pre = 'ability_encoder.mlp2'
        h2 = F.elu(F.linear(hid_mean, params[f'{pre}.0.weight'], params[f'{pre}.0.bias']))
        mu, logvar = torch.chunk(F.linear(h2, params[f'{pre}.2.weight'], params[f'{pre}.2.bias']), 2, dim=1)
        return mu, logvar
    out = encoder_mlp(params, x)
    mu_set, lv_set = torch.chunk(out, 2, dim=1)
    mu_set = mu_set.reshape(B, I, ability_dim).permute(1, 0, 2)   # [I,B,A]
    lv_set = lv_set.reshape(B, I, ability_dim).permute(1, 0, 2)
    # the reference's own host-side steps (models.py:597-606): a .item() sync on sum(1 - mask), two zero tensors for the
    # prior experts (allocated whether or not a cell is missing)
    has_missing = bool(torch.sum(1 - mask.long()).item()) if mask.dtype != torch.bool else not bool(mask.all())
    p_mu_set, p_lv_set = torch.zeros_like(mu_set), torch.zeros_like(lv_set)   # noqa: F841  (prior experts, models.py:603-604)
    if not has_missing:
        # nothing missing in the batch: the vectorised branch (models.py:626-627), no masking
        return product_of_experts(mu_set, lv_set)
    obs = mask.to(mu_set.dtype).t().unsqueeze(2)                    # [I,B,1]
    if replace_missing_with_prior:
        mu_set = mu_set * obs            # prior expert: mu 0
        lv_set = lv_set * obs            # prior expert: logvar 0
        return product_of_experts(mu_set, lv_set)
    return product_of_experts(mu_set, lv_set, weight=obs)


def irt_link(irt_model, ability, item_feat):
    """P(response = 1) for every (person, item): [B,I]."""
    A = ability.shape[1]
    if irt_model == 1:
        return torch.sigmoid(ability.sum(1, keepdim=True) + item_feat.t())
    logit = ability @ (-item_feat[:, :A].t()) + item_feat[:, A:A + 1].t()
    if irt_model == 2:
        return torch.sigmoid(logit)
    guess = torch.sigmoid(item_feat[:, A + 1:A + 2]).t()           # [1,I]
    return guess + (1.0 - guess) * torch.sigmoid(logit)


def irt_logit(irt_model, ability, item_feat):
    """irt_model_*pl(..., return_logit=True) (models.py:729-766): ([B,I] logit, 3PL guess [1,I] or None)."""
    A = ability.shape[1]
    if irt_model == 1:
        return ability.sum(1, keepdim=True) + item_feat.t(), None
    logit = ability @ (-item_feat[:, :A].t()) + item_feat[:, A:A + 1].t()
    guess = torch.sigmoid(item_feat[:, A + 1:A + 2]).t() if irt_model == 3 else None
    return logit, guess


def decoder_probs(params, generative_model, irt_model, ability, item_feat):
    """P(response = 1) [B,I] of the per-term MLP decoders (models.py:769-919), evaluated per (person, item) term the
    way the reference does: LinkedIRT.forward :783-801, DeepIRT.forward :851-866, ResidualIRT.forward :900-916."""
    def mlp(prefix, x, sigmoid=False):
        h = F.elu(F.linear(x, params[f'{prefix}.0.weight'], params[f'{prefix}.0.bias']))
        h = F.elu(F.linear(h, params[f'{prefix}.2.weight'], params[f'{prefix}.2.bias']))
        o = F.linear(h, params[f'{prefix}.4.weight'], params[f'{prefix}.4.bias'])
        return torch.sigmoid(o) if sigmoid else o
    B, I = ability.shape[0], item_feat.shape[0]
    if generative_model == 'link':
        logit, guess = irt_logit(irt_model, ability, item_feat)
        mu = mlp('decoder.link', logit.reshape(B * I, 1), sigmoid=True).reshape(B, I)
        return mu if guess is None else guess + (1.0 - guess) * mu
    hid_a = mlp('decoder.mlp_ability', ability).unsqueeze(1).expand(B, I, -1)
    hid_i = mlp('decoder.mlp_item_feat', item_feat).unsqueeze(0).expand(B, I, -1)
    res = mlp('decoder.mlp_concat', torch.cat([hid_i, hid_a], dim=2).reshape(B * I, -1)).reshape(B, I)
    if generative_model == 'deep':
        return torch.sigmoid(res)
    logit, guess = irt_logit(irt_model, ability, item_feat)
    mu = torch.sigmoid(res + logit)
    return mu if guess is None else guess + (1.0 - guess) * mu


def planar_flows(params, prefix, z, n_flows):
    """Sequence of planar flows; returns (z_K, sum of log|det J|) per row."""
    ladj = torch.zeros(z.shape[0], dtype=z.dtype)
    for k in range(n_flows):
        u = params[f'{prefix}.flows.{k}.u']
        w = params[f'{prefix}.flows.{k}.w']
        b = params[f'{prefix}.flows.{k}.b']
        uw = torch.dot(u, w)
        uhat = u + (F.softplus(uw) - 1.0 - uw) * w / torch.sum(w * w)
        t = torch.tanh(z @ w + b)
        z = z + uhat.unsqueeze(0) * t.unsqueeze(1)
        psi_u = (1.0 - t * t) * torch.dot(w, uhat)
        ladj = ladj + torch.log(torch.abs(1.0 + psi_u) + 1e-8)
    return z, ladj


def masked_bernoulli_ll(x, mask, probs):
    """Bernoulli(probs).log_prob(x) * mask with torch.distributions' numerics:
    probs are clamped to [eps, 1-eps] (eps = float32 machine eps), converted to
    logits, and scored with BCE-with-logits."""
    if probs.dtype == torch.float32:
        # the reference's own call (utils.py:46-49): same library path, same host-side cost per step
        return torch.distributions.Bernoulli(probs=probs, validate_args=False).log_prob(x) * mask.float()
    eps = torch.finfo(torch.float32).eps if probs.dtype == torch.float32 else torch.finfo(probs.dtype).eps
    pc = probs.clamp(min=eps, max=1.0 - eps)
    logits = torch.log(pc) - torch.log1p(-pc)
    return -F.binary_cross_entropy_with_logits(logits, x, reduction='none') * mask.to(probs.dtype)


def kl_std_normal(mu, logvar):
    return (-0.5 * (1.0 + logvar - mu * mu - logvar.exp())).sum(1)


def normal_logpdf(x, mu, logvar):
    return -0.5 * LOG_2PI - 0.5 * logvar - 0.5 * (x - mu) ** 2 / logvar.exp()


def std_normal_logpdf(x):
    return -0.5 * LOG_2PI - 0.5 * x * x


def elbo_forward(params, response, mask, eps_item, eps_ability, *, irt_model,
                 ability_dim, conditional_posterior=False,
                 replace_missing_with_prior=True, n_norm_flows=0,
                 annealing_factor=1.0, use_kl_divergence=True, generative_model='irt'):
    """One ELBO evaluation.  Returns a dict with ``loss`` (= -ELBO summed over
    the minibatch, models.py:443) and every intermediate forward() returns."""
    irt_model = int(irt_model)
    # ItemInferenceNetwork (models.py:713-726): both embeddings looked up for ALL items every step
    item_index = torch.arange(params['item_encoder.mu_lookup.weight'].shape[0])
    item_mu = F.embedding(item_index, params['item_encoder.mu_lookup.weight'])
    item_lv = F.embedding(item_index, params['item_encoder.logvar_lookup.weight'])
    item_feat = eps_item * torch.exp(0.5 * item_lv) + item_mu
    amu, alv = ability_posterior(
        params, response, mask, item_feat, ability_dim=ability_dim,
        conditional_posterior=conditional_posterior,
        replace_missing_with_prior=replace_missing_with_prior)
    ability = eps_ability * torch.exp(0.5 * alv) + amu

    out = dict(ability=ability, ability_mu=amu, ability_logvar=alv,
               item_feat=item_feat, item_feat_mu=item_mu, item_feat_logvar=item_lv)
    if n_norm_flows > 0:
        ability_k, a_ladj = planar_flows(params, 'ability_norm_flows', ability, n_norm_flows)
        item_k, i_ladj = planar_flows(params, 'item_norm_flows', item_feat, n_norm_flows)
        out.update(ability_k=ability_k, ability_logabsdetjac=a_ladj,
                   item_feat_k=item_k, item_feat_logabsdetjac=i_ladj)
        th, it = ability_k, item_k
    else:
        th, it = ability, item_feat
    probs = irt_link(irt_model, th, it) if generative_model == 'irt' else decoder_probs(params, generative_model, irt_model, th, it)
    out['response_mu'] = probs

    ll = masked_bernoulli_ll(response, mask, probs).sum()
    out['log_lik'] = ll
    if n_norm_flows > 0:
        log_q = (normal_logpdf(ability, amu, alv).sum() - a_ladj.sum()
                 + normal_logpdf(item_feat, item_mu, item_lv).sum() - i_ladj.sum())
        log_p = ll + std_normal_logpdf(ability_k).sum() + std_normal_logpdf(item_k).sum()
        elbo = log_p - log_q
    elif use_kl_divergence:
        kl_u = kl_std_normal(amu, alv).sum()
        kl_d = kl_std_normal(item_mu, item_lv).sum()
        out.update(kl_ability=kl_u, kl_item=kl_d)
        elbo = ll - annealing_factor * kl_u - annealing_factor * kl_d
    else:
        log_p = ll + std_normal_logpdf(ability).sum() + std_normal_logpdf(item_feat).sum()
        log_q = normal_logpdf(ability, amu, alv).sum() + normal_logpdf(item_feat, item_mu, item_lv).sum()
        elbo = log_p - log_q
    out['loss'] = -elbo
    return out


def vi_elbo_forward(params, index, response, mask, eps_item, eps_ability, *, irt_model, ability_dim,
                    annealing_factor=1.0, use_kl_divergence=True):
    """Un-amortized VI (models.py:100-243: VI_1PL/2PL/3PL.forward + elbo): per-person Gaussian posteriors looked up by
    `index` in two embeddings instead of an encoder; same link, log-likelihood and KL / sampled regulariser."""
    item_mu, item_lv = params['item_mu_lookup.weight'], params['item_logvar_lookup.weight']
    item_feat = eps_item * torch.exp(0.5 * item_lv) + item_mu                      # models.py:128-131
    amu, alv = params['ability_mu_lookup.weight'][index], params['ability_logvar_lookup.weight'][index]
    ability = eps_ability * torch.exp(0.5 * alv) + amu                             # models.py:133-135
    probs = irt_link(int(irt_model), ability, item_feat)
    ll = masked_bernoulli_ll(response, mask, probs).sum()
    if use_kl_divergence:                                                          # models.py:157-160
        elbo = ll - annealing_factor * kl_std_normal(amu, alv).sum() - annealing_factor * kl_std_normal(item_mu, item_lv).sum()
    else:                                                                          # models.py:161-170
        log_p = ll + std_normal_logpdf(ability).sum() + std_normal_logpdf(item_feat).sum()
        log_q = normal_logpdf(ability, amu, alv).sum() + normal_logpdf(item_feat, item_mu, item_lv).sum()
        elbo = log_p - log_q
    return dict(loss=-elbo, ability=ability, ability_mu=amu, ability_logvar=alv, item_feat=item_feat, response_mu=probs)


def elbo_loss_and_grads(params, response, mask, eps_item, eps_ability, **cfg):
    """loss + d loss / d every parameter (autograd), as float tensors."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    out = elbo_forward(leaves, response, mask, eps_item, eps_ability, **cfg)
    names = list(leaves)
    grads = torch.autograd.grad(out['loss'], [leaves[n] for n in names], allow_unused=True)
    gdict = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(names, grads)}
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, gdict


# ---------------------------------------------------------------------------
# parameter construction (models.py:281-329, 512-518; flows.py:17-19)
# ---------------------------------------------------------------------------

def init_params(irt_model, ability_dim, num_item, *, hidden_dim=64,
                conditional_posterior=False, n_norm_flows=0, generator=None,
                dtype=torch.float32, ability_merge='product'):
    """Random parameters with the reference's shapes and init distributions
    (xavier-normal(gain=sqrt 2) linears with zero bias, N(0,1) embeddings and
    flow u/w, flow b = 1).  NOT bit-identical to the reference's RNG order --
    goldens carry the reference's own state_dict."""
    g = generator
    D = item_feat_dim(irt_model, ability_dim)
    in_dim = 1 + (D if conditional_posterior else 0)
    dims = [(in_dim, hidden_dim), (hidden_dim, hidden_dim), (hidden_dim, 2 * ability_dim)]
    p = {}
    if ability_merge == 'mean':      # models.py:584-594
        names = [('mlp1.0', in_dim, hidden_dim), ('mlp1.2', hidden_dim, hidden_dim),
                 ('mlp2.0', hidden_dim, hidden_dim), ('mlp2.2', hidden_dim, 2 * ability_dim)]
    else:
        names = [(f'mlp.{idx}', fi, fo) for idx, (fi, fo) in zip((0, 2, 4), dims)]
    for name, fi, fo in names:
        std = math.sqrt(2.0) * math.sqrt(2.0 / (fi + fo))
        p[f'ability_encoder.{name}.weight'] = torch.randn(fo, fi, generator=g, dtype=dtype) * std
        p[f'ability_encoder.{name}.bias'] = torch.zeros(fo, dtype=dtype)
    p['item_encoder.mu_lookup.weight'] = torch.randn(num_item, D, generator=g, dtype=dtype)
    p['item_encoder.logvar_lookup.weight'] = torch.randn(num_item, D, generator=g, dtype=dtype)
    for name, dim in (('ability_norm_flows', ability_dim), ('item_norm_flows', D)):
        for k in range(n_norm_flows):
            p[f'{name}.flows.{k}.u'] = torch.randn(dim, generator=g, dtype=dtype)
            p[f'{name}.flows.{k}.w'] = torch.randn(dim, generator=g, dtype=dtype)
            p[f'{name}.flows.{k}.b'] = torch.ones(1, dtype=dtype)
    return p


def simulate_responses(irt_model, num_person, num_item, ability_dim, generator=None,
                       missing_frac=0.0):
    """Synthetic responses with the semantics of the reference generator
    (src/pyro_core/models.py:25-159 via src/simulate.py:40-58): theta ~ N(0,1),
    item ~ N(0,1), r ~ Bernoulli(link).  Returns response [P,I] float (-1 where
    masked), mask [P,I] uint8."""
    g = generator
    D = item_feat_dim(irt_model, ability_dim)
    theta = torch.randn(num_person, ability_dim, generator=g)
    item = torch.randn(num_item, D, generator=g)
    probs = irt_link(int(irt_model), theta, item)
    resp = torch.bernoulli(probs, generator=g)
    mask = torch.ones(num_person, num_item, dtype=torch.uint8)
    if missing_frac > 0:
        drop = torch.rand(num_person, num_item, generator=g) < missing_frac
        resp = torch.where(drop, torch.full_like(resp, -1.0), resp)
        mask = (~drop).to(torch.uint8)
    return resp, mask
