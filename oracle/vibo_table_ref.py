"""TEST INFRASTRUCTURE ONLY -- analytic ("table") restatement of the fused ELBO.

The HIP kernel does not run the encoder MLP per (person,item): Bernoulli
responses take two observed values, so the MLP collapses to a lookup table
``E[c] = MLP([c])`` (unconditional posterior, c in {0,1}) or
``E[c,i] = MLP([c, d_i])`` (conditional posterior); missing cells are the fixed
N(0,1) prior expert or dropped (models.py:596-629).  This file restates the
fused forward + hand-derived backward the kernel implements, on CPU with plain
tensor ops (NO autograd), with exactly the kernel's inputs/outputs:

    inputs : table, item sample (or item_K), response/mask, eps, flow params
    outputs: heads  ll   = sum masked Bernoulli log-lik
                    reg  = sum_p KL(q(theta_p)||N(0,1))                 (mode 'kl')
                         = sum_p [log q0(theta_0) - ladj - log p(theta_K)] (mode 'sampled')
             ability_mu / ability_logvar / ability (/ ability_k, ladj)
             g_table[2] (d ll / d table, d reg / d table), g_item (d ll / d item),
             g_flow[2]

It is checked (tests/test_host_logic.py::test_table_ref_matches_autograd_oracle) against autograd through
oracle/vibo_oracle.py, which is itself pinned to the reference goldens; it is
then the small-shape checker for the kernel's raw outputs and the stand-in the
CPU host-logic tests monkeypatch in place of the C-ABI call.

Reference lines: models.py:356-371 (encode), 596-629 (PoE + missing), 729-766
(links), 380-443 (elbo), flows.py:21-41, utils.py:46-49,59-67,85-88,105-113.
"""
import math

import torch

LOG_2PI = math.log(2.0 * math.pi)
EPS32 = 1.1920928955078125e-07            # torch.finfo(float32).eps
LOGIT_LO = 15.942384719848633             # -log(eps32/(1-eps32)): p < eps below -LOGIT_LO
LOGIT_HI = 16.635532333438686             # 24 ln 2: sigmoid(l) rounds to 1.0f above


def encoder_table(params, item_feat=None, conditional_posterior=False, prefix='ability_encoder.mlp'):
    """[2,2A] (uncond) or [2,I,2A] (cond) table of encoder outputs for the two
    observed response values 0 and 1.  Built with torch ops so autograd can
    carry table gradients back into the MLP (and into item_feat when cond)."""
    import torch.nn.functional as F
    w0, b0 = params[f'{prefix}.0.weight'], params[f'{prefix}.0.bias']
    vals = torch.tensor([[0.0], [1.0]], dtype=w0.dtype)
    if conditional_posterior:
        I = item_feat.shape[0]
        x = torch.cat([vals.unsqueeze(1).expand(2, I, 1), item_feat.unsqueeze(0).expand(2, I, -1)], dim=2)
    else:
        x = vals
    h = F.elu(F.linear(x, w0, b0))
    h = F.elu(F.linear(h, params[f'{prefix}.2.weight'], params[f'{prefix}.2.bias']))
    return F.linear(h, params[f'{prefix}.4.weight'], params[f'{prefix}.4.bias'])


def flow_uhat(u, w):
    """flows.py:23-25 (autograd-able; done on the host, not in the kernel)."""
    import torch.nn.functional as F
    uw = torch.dot(u, w)
    return u + (F.softplus(uw) - 1.0 - uw) * w / torch.sum(w * w)


def fused_elbo_ref(table, item, response, mask, eps, *, irt_model, ability_dim,
                   conditional_posterior=False, replace_missing_with_prior=True,
                   mode='kl', flow_uhat_w_b=None, want_grad=True, exact_saturation=True, given_posterior=False):
    """See module docstring.  table [2,2A] | [2,I,2A]; item [I,D]; response
    [B,I] (1.0 = correct); mask [B,I] (nonzero = observed); eps [B,A];
    flow_uhat_w_b: list of (uhat[A], w[A], b[1]) or None."""
    dt = table.dtype
    A = ability_dim
    B, I = response.shape
    k = (mask != 0).to(dt)                                   # [B,I]
    x = (response == 1).to(dt)
    irt_model = int(irt_model)

    # ---- caller-supplied posterior (VIBO_POSTERIOR_GIVEN: table = [B,2A] mu | logvar; --ability-merge mean) ----
    if given_posterior:
        m = s = tau_obs = None
        k3 = k.unsqueeze(2)
    # ---- product of experts -------------------------------------------------
    elif conditional_posterior:
        m_tab, s_tab = table[..., :A], table[..., A:]        # [2,I,A]
        c = x.long()                                         # [B,I]
        idx = torch.arange(I).unsqueeze(0).expand(B, I)
        m = m_tab[c, idx]                                    # [B,I,A]
        s = s_tab[c, idx]
    else:
        m_tab, s_tab = table[:, :A], table[:, A:]            # [2,A]
        m = m_tab[x.long()]                                  # [B,I,A]
        s = s_tab[x.long()]
    if given_posterior:
        amu, alv = table[:, :A], table[:, A:]
        lam = torch.exp(-alv)
    else:
        tau_obs = 1.0 / (torch.exp(s) + 1e-8)
        k3 = k.unsqueeze(2)
        if replace_missing_with_prior:
            tau_prior = 1.0 / (1.0 + 1e-8)
            tau = k3 * tau_obs + (1.0 - k3) * tau_prior
            m_eff = k3 * m
        else:
            tau = k3 * tau_obs
            m_eff = m
        lam = tau.sum(1)                                         # [B,A]
        amu = (m_eff * tau).sum(1) / lam
        alv = torch.log(1.0 / lam)
    sig = torch.exp(0.5 * alv)
    theta0 = eps * sig + amu

    # ---- planar flows on the ability sample --------------------------------
    flows = flow_uhat_w_b or []
    z = theta0
    ladj = torch.zeros(B, dtype=dt)
    saved = []
    for (uhat, w, b) in flows:
        a = z @ w + b
        t = torch.tanh(a)
        cwu = torch.dot(w, uhat)
        one_psi = 1.0 + (1.0 - t * t) * cwu
        ladj = ladj + torch.log(one_psi.abs() + 1e-8)
        saved.append((z, t, cwu, one_psi))
        z = z + uhat.unsqueeze(0) * t.unsqueeze(1)
    theta = z

    # ---- link + masked Bernoulli log-lik ------------------------------------
    if irt_model == 1:
        logit = theta.sum(1, keepdim=True) + item[:, 0].unsqueeze(0)
    else:
        logit = -(theta @ item[:, :A].t()) + item[:, A].unsqueeze(0)
    if exact_saturation:
        lc = logit.clamp(-LOGIT_LO, LOGIT_LO)
        live = ((logit >= -LOGIT_LO) & (logit <= LOGIT_HI)).to(dt)
    else:
        lc, live = logit, torch.ones_like(logit)
    sgm = torch.sigmoid(lc)
    if irt_model == 3:
        guess = torch.sigmoid(item[:, A + 1]).unsqueeze(0)   # [1,I]
        p = guess + (1.0 - guess) * sgm
        pc = p.clamp(EPS32, 1.0 - EPS32)
        live3 = ((p >= EPS32) & (p <= 1.0 - EPS32)).to(dt) * live
        ll_t = x * torch.log(pc) + (1.0 - x) * torch.log1p(-pc)
        dll_dp = (x / pc - (1.0 - x) / (1.0 - pc)) * live3
        gl = k * dll_dp * (1.0 - guess) * sgm * (1.0 - sgm)            # d ll / d logit
        gguess = k * dll_dp * (1.0 - sgm) * guess * (1.0 - guess)      # d ll / d guess-logit
    else:
        # log sigmoid(+-l) = x*l - softplus(l)
        ll_t = x * lc - torch.clamp(lc, min=0) - torch.log1p(torch.exp(-lc.abs()))
        gl = k * (x - sgm) * live
    ll = (k * ll_t).sum()

    out = dict(ll=ll, ability_mu=amu, ability_logvar=alv, ability=theta0,
               ability_k=theta, ladj=ladj, logit=logit)
    kl_u = (-0.5 * (1.0 + alv - amu * amu - alv.exp())).sum()
    logq0 = (-0.5 * LOG_2PI - 0.5 * alv - 0.5 * eps * eps).sum()
    logp = (-0.5 * LOG_2PI - 0.5 * theta * theta).sum()
    out.update(kl_ability=kl_u, logq0=logq0, logp=logp, ladj_sum=ladj.sum())
    if mode == 'kl':
        assert not flows
        out['reg'] = kl_u
    else:
        out['reg'] = logq0 - ladj.sum() - logp
    if not want_grad:
        return out

    # ---- backward -----------------------------------------------------------
    # d ll / d item
    D = item.shape[1]
    g_item = torch.zeros(I, D, dtype=dt)
    if irt_model == 1:
        g_item[:, 0] = gl.sum(0)
        gth_ll = gl.sum(1, keepdim=True).expand(B, A).clone()
    else:
        g_item[:, :A] = -(gl.t() @ theta)
        g_item[:, A] = gl.sum(0)
        gth_ll = -(gl @ item[:, :A])
        if irt_model == 3:
            g_item[:, A + 1] = gguess.sum(0)
    gth_reg = theta.clone() if mode != 'kl' else torch.zeros_like(theta)
    gladj_reg = -1.0 if mode != 'kl' else 0.0

    g_flow = [[], []]
    gz_sets = [gth_ll, gth_reg]
    gl_sets = [0.0, gladj_reg]
    for s_idx in range(2):
        gz = gz_sets[s_idx]
        gla = gl_sets[s_idx]
        per_flow = []
        for (uhat, w, b), (zin, t, cwu, one_psi) in zip(reversed(flows), reversed(saved)):
            dl_dpsi = gla * torch.sign(one_psi) / (one_psi.abs() + 1e-8)        # [B]
            g_t = gz @ uhat + dl_dpsi * (-2.0 * t * cwu)
            g_c = (dl_dpsi * (1.0 - t * t)).sum()
            g_a = g_t * (1.0 - t * t)
            g_uhat = (gz * t.unsqueeze(1)).sum(0) + g_c * w
            g_w = (g_a.unsqueeze(1) * zin).sum(0) + g_c * uhat
            g_b = g_a.sum().reshape(1)
            gz = gz + g_a.unsqueeze(1) * w.unsqueeze(0)
            per_flow.append((g_uhat, g_w, g_b))
        g_flow[s_idx] = list(reversed(per_flow))
        gz_sets[s_idx] = gz
    g_mu = [gz_sets[0].clone(), gz_sets[1].clone()]
    g_lv = [gz_sets[0] * 0.5 * sig * eps, gz_sets[1] * 0.5 * sig * eps]
    if mode == 'kl':
        g_mu[1] = g_mu[1] + amu
        g_lv[1] = g_lv[1] - 0.5 * (1.0 - alv.exp())
    else:
        g_lv[1] = g_lv[1] - 0.5

    if given_posterior:
        out.update(g_table=[torch.cat([g_mu[i], g_lv[i]], dim=1) for i in range(2)], g_item=g_item, g_flow=g_flow)
        return out
    g_table = []
    for s_idx in range(2):
        gm_p = (g_mu[s_idx] / lam).unsqueeze(1)                       # [B,1,A]
        glv_p = (g_lv[s_idx] / lam).unsqueeze(1)
        g_m = k3 * gm_p * tau_obs                                     # d/d m_pi (observed only)
        g_tau = gm_p * (m - amu.unsqueeze(1)) - glv_p
        g_s = -k3 * g_tau * tau_obs * tau_obs * torch.exp(s)
        gt = torch.zeros_like(table)
        for c_val in (0, 1):
            sel = (x == c_val).to(dt).unsqueeze(2)                    # [B,I,1]
            if conditional_posterior:
                gt[c_val, :, :A] = (g_m * sel).sum(0)
                gt[c_val, :, A:] = (g_s * sel).sum(0)
            else:
                gt[c_val, :A] = (g_m * sel).sum((0, 1))
                gt[c_val, A:] = (g_s * sel).sum((0, 1))
        g_table.append(gt)
    out.update(g_table=g_table, g_item=g_item, g_flow=g_flow)
    return out
