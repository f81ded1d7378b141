"""The per-term MLP decoder kernel (vibo_decoder_fwd_bwd: --generative-model link | deep | residual, models.py:769-919)
against a plain PyTorch float64 statement of the same network with autograd."""
import pytest
import torch
import torch.nn.functional as F

from vibo_amd import decoder as D

pytestmark = pytest.mark.gpu
EPS32 = 1.1920928955078125e-07


def torch_reference(resp, mask, U, V, W2, b2, w3, b3, logit, w1, guess, resid):
    """[B, I, 64] the slow way (float64)."""
    z1 = V.unsqueeze(1) + (U.unsqueeze(0) if U is not None else 0.0)
    if w1 is not None:
        z1 = z1 + logit.unsqueeze(2) * w1
    h2 = F.elu(F.elu(z1) @ W2.t() + b2)
    o = h2 @ w3 + b3
    if resid:
        o = o + resid * logit
    p = torch.sigmoid(o)
    if guess is not None:
        p = guess + (1 - guess) * p
    pc = p.clamp(EPS32, 1 - EPS32)                       # torch.distributions.Bernoulli(probs=...) clamp (utils.py:46-49)
    ll = torch.where(resp > 0.5, pc.log(), torch.log1p(-pc))
    if mask is not None:
        ll = ll * mask
    return ll.sum(), p


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('mode,B,I,missing,H', [('deep', 37, 130, 0.2, 64), ('residual', 16, 64, 0.0, 64), ('link', 33, 95, 0.3, 64),
                                               ('residual3', 50, 200, 0.1, 64), ('link3', 7, 20, 0.0, 64), ('deep', 300, 1000, 0.1, 64),
                                               # --hidden-dim below 64 (vibo.py:72; models.py:771-781, 815-846): the same kernel on
                                               # zero-padded weights, the padding sliced off the gradients again
                                               ('deep', 37, 130, 0.2, 32), ('link3', 21, 95, 0.1, 16), ('residual', 40, 64, 0.2, 48)])
def test_decoder_kernel_matches_float64_autograd(mode, B, I, missing, H):
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B * 1000 + I)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).double()
    resp = (torch.rand(B, I, generator=g) < 0.5).double()
    mask = (torch.rand(B, I, generator=g) >= missing).double() if missing > 0 else None
    t = dict(V=rn(B, H, sc=0.7), W2=rn(H, H, sc=0.18), b2=rn(H, sc=0.1), w3=rn(H, sc=0.25), b3=rn(1, sc=0.1))
    has_l = mode != 'deep'
    t['U'] = rn(I, H, sc=0.7) if not mode.startswith('link') else None
    t['logit'] = rn(B, I, sc=1.5) if has_l else None
    t['w1'] = rn(H, sc=0.5) if mode.startswith('link') else None
    t['guess'] = torch.sigmoid(rn(I)) if mode.endswith('3') else None
    resid = 1.0 if mode.startswith('residual') else 0.0
    leaves = {k: v.clone().requires_grad_(True) for k, v in t.items() if v is not None}
    ll_ref, p_ref = torch_reference(resp, mask, leaves.get('U'), leaves['V'], leaves['W2'], leaves['b2'], leaves['w3'], leaves['b3'],
                                    leaves.get('logit'), leaves.get('w1'), leaves.get('guess'), resid)
    ll_ref.backward()

    dl = {k: v.float().to(dev).requires_grad_(True) for k, v in t.items() if v is not None}
    args = dict(U=dl.get('U'), V=dl['V'], W2=dl['W2'], b2=dl['b2'], w3=dl['w3'], b3=dl['b3'], logit=dl.get('logit'),
                w1=dl.get('w1'), guess=dl.get('guess'), resid=resid)
    r32 = resp.float().to(dev)
    m8 = mask.bool().to(dev) if mask is not None else None
    ll = D.decoder_log_lik(r32, m8, **args)
    ll.backward()
    assert abs(float(ll.detach()) - float(ll_ref.detach())) < 2e-5 * abs(float(ll_ref.detach()))
    for k in leaves:
        assert rel(dl[k].grad.double().cpu(), leaves[k].grad) < 2e-4, k
    p = D.decoder_probs(B, I, **{k: (v.detach() if torch.is_tensor(v) else v) for k, v in args.items()})
    assert (p.double().cpu() - p_ref.detach()).abs().max() < 2e-5
    # fixed-order partial records: bitwise reproducible
    ll2 = D.decoder_log_lik(r32, m8, **args)
    assert torch.equal(ll.detach(), ll2.detach())


def test_person_chunking_changes_nothing_but_the_summation_order(monkeypatch):
    """Calls with more persons than decoder.PERSON_CHUNK go through in several launches (bounded scratch); same sums."""
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(3)
    B, I, H = 700, 130, 64
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).requires_grad_(True)
    resp = (torch.rand(B, I, device=dev, generator=g) < 0.5).float()
    mask = torch.rand(B, I, device=dev, generator=g) >= 0.2
    t = dict(U=rn(I, H, sc=0.7), V=rn(B, H, sc=0.7), W2=rn(H, H, sc=0.18), b2=rn(H, sc=0.1), w3=rn(H, sc=0.25), b3=rn(1, sc=0.1),
             logit=rn(B, I, sc=1.5), guess=torch.sigmoid(torch.randn(I, device=dev, generator=g)).requires_grad_(True), resid=1.0)
    res = []
    for chunk in (1 << 20, 256):
        monkeypatch.setattr(D, 'PERSON_CHUNK', chunk)
        for v in t.values():
            if torch.is_tensor(v):
                v.grad = None
        ll = D.decoder_log_lik(resp, mask, **t)
        ll.backward()
        p = D.decoder_probs(B, I, **{k: (v.detach() if torch.is_tensor(v) else v) for k, v in t.items()})
        res.append((ll.detach().clone(), {k: v.grad.clone() for k, v in t.items() if torch.is_tensor(v)}, p))
    (l0, g0, p0), (l1, g1, p1) = res
    assert abs(float(l0) - float(l1)) < 1e-6 * abs(float(l0)) and torch.equal(p0, p1)
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-6 * float(g0[k].abs().max()), k
    assert torch.equal(g0['V'], g1['V']) and torch.equal(g0['logit'], g1['logit'])       # per-person outputs: bitwise


@pytest.mark.parametrize('N,Dm,K', [(1000, 3, 4), (37, 10, 8), (10000, 2, 1), (257, 1, 2)])
def test_flow_stack_kernels_match_autograd(N, Dm, K):
    """vibo_flow_stack_forward / _backward (the item-side planar-flow stack, flows.py:21-66) vs float64 autograd."""
    from vibo_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(N + Dm + K)
    z = torch.randn(N, Dm, generator=g).double().requires_grad_(True)
    u, w = torch.randn(K, Dm, generator=g).double(), torch.randn(K, Dm, generator=g).double()
    uw = (u * w).sum(1, keepdim=True)          # uhat as the flow builds it (flows.py:24-26): w . uhat >= -1, so psi stays away from 0
    uhat = u + (torch.nn.functional.softplus(uw) - 1.0 - uw) * w / (w * w).sum(1, keepdim=True)
    packed = torch.cat([uhat, w, torch.randn(K, 1, generator=g).double()], 1).requires_grad_(True)
    zz, total = z, 0.0
    for k in range(K):
        uhat, w, b = packed[k, :Dm], packed[k, Dm:2 * Dm], packed[k, 2 * Dm]
        t = torch.tanh(zz @ w + b)
        zz = zz + uhat.unsqueeze(0) * t.unsqueeze(1)
        total = total + torch.log(torch.abs(1.0 + (1.0 - t * t) * torch.dot(w, uhat)) + 1e-8)
    cz, cl = torch.randn(N, Dm, generator=g).double(), torch.randn(N, generator=g).double()
    ((zz * cz).sum() + (total * cl).sum()).backward()
    z32 = z.detach().float().to(dev).requires_grad_(True)
    p32 = packed.detach().float().to(dev).requires_grad_(True)
    zo, la = ops.FlowStackFn.apply(z32, p32)
    ((zo * cz.float().to(dev)).sum() + (la * cl.float().to(dev)).sum()).backward()
    assert (zo.detach().double().cpu() - zz.detach()).abs().max() < 2e-6 * max(1.0, float(zz.abs().max()))
    assert (la.detach().double().cpu() - total.detach()).abs().max() < 2e-5
    assert rel(z32.grad.double().cpu(), z.grad) < 1e-4
    assert rel(p32.grad.double().cpu(), packed.grad) < 1e-4
    zo2, la2 = ops.FlowStackFn.apply(z32, p32)
    assert torch.equal(zo, zo2) and torch.equal(la, la2)


def test_decoder_kernel_on_the_references_saturation_probe():
    """The Bernoulli probability clamp of the reference (utils.py:46-49 -> torch: value capped, gradient exactly zero outside the
    clamp and passed AT its bounds) through the decoder kernel: residual mode with a zero network makes the output equal the
    supplied logit, and tests/golden/saturation.npz holds the REAL reference's ll and d ll / d logit on a grid over [-30, 30]."""
    import os
    import numpy as np
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'saturation.npz'))
    dev = torch.device('cuda:0')
    logit = torch.from_numpy(z['logit'])[::3].to(dev)
    I, H = logit.numel(), 64
    zeros = lambda *s: torch.zeros(*s, device=dev)
    for x in (0, 1):
        L = logit.unsqueeze(0).clone().requires_grad_(True)
        ll = D.decoder_log_lik(torch.full((1, I), float(x), device=dev), None, U=zeros(I, H), V=zeros(1, H).requires_grad_(True),
                               W2=zeros(H, H), b2=zeros(H), w3=zeros(H), b3=zeros(1), logit=L, resid=1.0)
        ll.backward()
        ref_ll = torch.from_numpy(z[f'll_x{x}'])[::3].double()
        ref_g = torch.from_numpy(z[f'dll_dlogit_x{x}'])[::3]
        g = L.grad[0].cpu()
        assert torch.equal(g == 0, ref_g == 0)                      # the same cells are clamped
        assert (g - ref_g).abs().max() < 2e-6
        # (the reference's own fp32 value is quantised through 1 - P for logits of 12..16 -- up to 5e-4 per cell; the kernel forms
        #  1 - P without that cancellation, so the sums agree to 1e-4, not to rounding)
        assert abs(float(ll.detach()) - float(ref_ll.sum())) < 1e-4 * abs(float(ref_ll.sum()))


@pytest.mark.parametrize('A,I,B,prior,codes', [(2, 95, 77, True, False), (8, 1000, 300, True, True), (3, 260, 64, False, False)])
def test_decoder_models_conditional_posterior_runs_on_the_code_table_sum_kernels(A, I, B, prior, codes):
    """--conditional-posterior with an MLP decoder (models.py:695-710 product of experts, utils.py:105-113): the experts'
    per-person sums come from vibo_code_table_sum_forward / _backward (one-hot x [tau | mu tau] on the matrix pipe) instead of a
    gathered [B, I, 2A] tensor -- posterior and the gradients that reach the encoder MLP and the item sample against the
    dense float64 statement of the same product."""
    from oracle import vibo_oracle as O
    from vibo_amd import ops
    from vibo_amd.torch_core.models import VIBO_2PL
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(A * 100 + I)
    resp, mask = O.simulate_responses(2, B, I, A, generator=g, missing_frac=0.2)
    torch.manual_seed(3)
    model = VIBO_2PL(A, I, ability_merge='product', conditional_posterior=True, generative_model='deep', replace_missing_with_prior=prior).to(dev)
    item_feat = torch.randn(I, A + 1, generator=g).to(dev).requires_grad_(True)
    r_in, m_in = (ops.pack_cell_codes(resp.to(dev), mask.bool().to(dev)), None) if codes else (resp.to(dev), mask.bool().to(dev))
    amu, alv = model._conditional_posterior_poe(r_in, m_in, None, item_feat)
    w = torch.randn(B, 2 * A, generator=g).to(dev)
    (torch.cat([amu, alv], 1) * w).sum().backward()
    got = [item_feat.grad.double().cpu()] + [p.grad.double().cpu() for p in model.ability_encoder.parameters()]
    # dense float64 reference
    m64 = __import__('copy').deepcopy(model).double()
    it64 = item_feat.detach().double().requires_grad_(True)
    table = m64.ability_encoder.expert_table(it64)                          # [2, I, 2A]
    r, m = resp.to(dev).double(), mask.to(dev).double()
    sel = table[(r == 1).long(), torch.arange(I, device=dev)]               # [B, I, 2A]
    tau = m.unsqueeze(2) / (torch.exp(sel[..., A:]) + 1e-8)
    lam = tau.sum(1) + ((I - m.sum(1, keepdim=True)) * (1.0 / (1.0 + 1e-8)) if prior else 0.0)
    amu_r, alv_r = (sel[..., :A] * tau).sum(1) / lam, torch.log(1.0 / lam)
    (torch.cat([amu_r, alv_r], 1) * w.double()).sum().backward()
    ref = [it64.grad.cpu()] + [p.grad.cpu() for p in m64.ability_encoder.parameters()]
    assert (amu.detach().double() - amu_r.detach()).abs().max() < 2e-5 and (alv.detach().double() - alv_r.detach()).abs().max() < 2e-5
    for a_, b_ in zip(got, ref):
        assert rel(a_, b_) < 2e-4
