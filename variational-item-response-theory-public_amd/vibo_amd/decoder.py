"""The per-term MLP decoders of --generative-model link | deep | residual (reference models.py:769-919) on the HIP
library's matrix-pipe kernel (csrc/vibo_decoder.hip, vibo_decoder_fwd_bwd).

All three decoders end in the same per-(person, item) network 64 -> ELU -> [64 x 64] -> ELU -> 64 -> 1.  The first
layer's input is a concatenation / a scalar, so it splits exactly into a per-item half U [I, 64], a per-person half
V [B, 64] and (link) a rank-one term w1 * logit; those halves -- a few small dense layers over I or B rows -- are
ordinary PyTorch modules here, and `decoder_log_lik` is the O(B I 64^2) rest: forward, masked Bernoulli log-likelihood
and the whole backward in one launch, returned to autograd as gradients of (U, V, logit, guess, w1, W2, b2, w3, b3).
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops

HIDDEN = 64


def _launch(response, mask, U, V, L, guess, w1, W2, b2, w3, b3, resid, want_grad, want_prob=False):
    lib = _lib.load()
    for t in (response, V, W2, b2, w3, b3):
        if t.device.type != 'cuda':
            raise _lib.ViboLibraryError('decoder_log_lik: device tensors only (no CPU fallback)')
    B, I = response.shape
    dev = response.device
    chunks = lib.vibo_decoder_person_chunks(B, I)
    n_ib = (I + 63) // 64
    n_wave = 4 * n_ib * chunks
    d = _lib.ViboDecoderDesc(B, I, HIDDEN, int(want_grad), chunks, float(resid), response.stride(0),
                             mask.stride(0) if mask is not None else 0)
    f32 = dict(dtype=torch.float32, device=dev)
    out = {'ll_part': torch.empty(n_wave, **f32)}
    if want_grad:
        out['dW2'] = torch.empty(n_wave, HIDDEN, HIDDEN, **f32)
        out['dvec'] = torch.empty(n_wave, 4, HIDDEN, **f32)
        out['dV'] = torch.empty(4 * n_ib, B, HIDDEN, **f32)
        if U is not None:
            out['dU'] = torch.empty(chunks, I, HIDDEN, **f32)
        if L is not None:
            out['dL'] = torch.empty(B, I, **f32)
        if guess is not None:
            out['dguess'] = torch.empty(chunks, I, **f32)
    if want_prob:
        out['prob'] = torch.empty(B, I, **f32)
    p = ops._ptr
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.vibo_decoder_fwd_bwd(ctypes.byref(d), p(response), p(mask), p(U), p(V), p(L), p(guess), p(w1), p(W2), p(b2), p(w3),
                                  p(b3), p(out['ll_part']), p(out.get('dU')), p(out.get('dV')), p(out.get('dL')),
                                  p(out.get('dguess')), p(out.get('dW2')), p(out.get('dvec')), p(out.get('prob')), stream)
    _lib.check(rc, 'vibo_decoder_fwd_bwd')
    return out


ops._BACKEND.setdefault('decoder', _launch)      # (tests without a GPU swap in oracle/cpu_backend.py's stand-in)


def _prep(t):
    return None if t is None else t.detach().contiguous().float()


PERSON_CHUNK = 65536      # persons per launch: bounds the per-wave d V records ([4 ceil(I / 64)][persons][64] floats) to ~1 GB at 1 000 items


class _DecoderLogLik(torch.autograd.Function):
    """sum of the masked Bernoulli log-likelihood; every gradient is produced by the forward launch(es) -- the persons go
    through in chunks of PERSON_CHUNK, the per-chunk partial records are summed here (fixed order)."""

    @staticmethod
    def forward(ctx, response, mask, resid, want_grad, U, V, L, guess, w1, W2, b2, w3, b3):
        # (want_grad is decided by the caller: inside Function.forward grad mode is off and nn.Parameters always say
        #  requires_grad -- under no_grad (test epochs, log_marginal's hundreds of samples) the GRAD kernel variant and its
        #  ~1 GB of d V records per 65 536 persons were produced for nothing)
        Up, Vp, Lp, gp, w1p, W2p, b2p, w3p, b3p = (_prep(t) for t in (U, V, L, guess, w1, W2, b2, w3, b3))
        B = response.shape[0]
        ll = None
        acc = {}
        for s in range(0, B, PERSON_CHUNK):
            e = min(B, s + PERSON_CHUNK)
            whole = s == 0 and e == B
            out = ops._BACKEND['decoder'](response if whole else response[s:e], mask if (mask is None or whole) else mask[s:e], Up,
                                          Vp if whole else Vp[s:e], Lp if (Lp is None or whole) else Lp[s:e], gp, w1p, W2p, b2p, w3p, b3p,
                                          resid, want_grad)
            part = out['ll_part'].sum()
            ll = part if ll is None else ll + part
            if want_grad:
                red = {'dW2': out['dW2'].sum(0), 'dvec': out['dvec'].sum(0)}
                if 'dU' in out:
                    red['dU'] = out['dU'].sum(0)
                if 'dguess' in out:
                    red['dguess'] = out['dguess'].sum(0)
                for k, v in red.items():
                    acc[k] = v if k not in acc else acc[k] + v
                acc.setdefault('dV', []).append(out['dV'].sum(0))
                if 'dL' in out:
                    acc.setdefault('dL', []).append(out['dL'])
        if want_grad:
            acc['dV'] = acc['dV'][0] if len(acc['dV']) == 1 else torch.cat(acc['dV'], 0)
            if 'dL' in acc:
                acc['dL'] = acc['dL'][0] if len(acc['dL']) == 1 else torch.cat(acc['dL'], 0)
        ctx.out = acc if want_grad else None
        ctx.has = (U is not None, L is not None, guess is not None, w1 is not None)
        return ll

    @staticmethod
    def backward(ctx, g):
        o = ctx.out
        has_u, has_l, has_g, has_w1 = ctx.has
        dvec = o['dvec']
        return (None, None, None, None,
                g * o['dU'] if has_u else None,
                g * o['dV'],
                g * o['dL'] if has_l else None,
                g * o['dguess'] if has_g else None,
                g * dvec[2] if has_w1 else None,
                g * o['dW2'], g * dvec[0], g * dvec[1], g * dvec[3, :1])


def _rows(response, mask):
    if isinstance(response, ops.CellCodes):
        response, mask = response.unpack()
    response = ops.prepare_response(response)
    if mask is not None:
        mask = ops.prepare_mask(mask)[0]
    return response.float(), mask


def _pad_hidden(U, V, W2, b2, w3, w1):
    """The kernel's per-term network is 64 units wide.  A narrower one (--hidden-dim < 64) is the same network with its extra
    units switched off: zero rows / columns give z1 = 0 -> elu(0) = 0 -> no contribution to the second layer, whose padded
    units see a zero bias and a zero output weight -- exact, and autograd slices the padding off the gradients again."""
    H = W2.shape[0]
    if H == HIDDEN:
        return U, V, W2, b2, w3, w1
    if H > HIDDEN:
        raise NotImplementedError(f'hidden_dim = {H}: the per-term decoder kernel covers hidden widths up to {HIDDEN} '
                                  '(its 64 x 64 weight-gradient tile lives in registers)')
    k = HIDDEN - H
    pad1 = lambda t: None if t is None else F.pad(t, (0, k))
    return pad1(U), pad1(V), F.pad(W2, (0, k, 0, k)), pad1(b2), pad1(w3), pad1(w1)


def decoder_log_lik(response, mask, *, U, V, W2, b2, w3, b3, logit=None, w1=None, guess=None, resid=0.0):
    """sum_{p,i} mask * log Bernoulli(response | P) of the per-term network (see the module docstring); differentiable in
    every tensor argument but response / mask.  response [B, I(, 1)] fp32, mask [B, I(, 1)] bool/u8 or None."""
    response, mask = _rows(response, mask)
    U, V, W2, b2, w3, w1 = _pad_hidden(U, V, W2, b2, w3, w1)
    want_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (U, V, logit, guess, w1, W2, b2, w3, b3))
    return _DecoderLogLik.apply(response, mask, float(resid), want_grad, U, V, logit, guess, w1, W2, b2, w3, b3)


@torch.no_grad()
def decoder_probs(B, I, *, U, V, W2, b2, w3, b3, logit=None, w1=None, guess=None, resid=0.0):
    """P(response = 1) [B, I] of the per-term network (decode(): models.py:373-378 with a non-IRT generative model)."""
    U, V, W2, b2, w3, w1 = _pad_hidden(U, V, W2, b2, w3, w1)
    Up, Vp, Lp, gp, w1p, W2p, b2p, w3p, b3p = (_prep(t) for t in (U, V, logit, guess, w1, W2, b2, w3, b3))
    outs = []
    for s in range(0, B, PERSON_CHUNK):
        e = min(B, s + PERSON_CHUNK)
        dummy = torch.zeros(e - s, I, device=V.device)
        out = ops._BACKEND['decoder'](dummy, None, Up, Vp[s:e], None if Lp is None else Lp[s:e], gp, w1p, W2p, b2p, w3p, b3p,
                                      resid, False, want_prob=True)
        outs.append(out['prob'])
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def irt_logit(irt_model, ability, item_feat):
    """irt_model_*pl(..., return_logit=True) (models.py:729-766): [B, I] logits (and the 3PL guess probability [I])."""
    A = ability.shape[1]
    if irt_model == 1:
        return ability.sum(1, keepdim=True) + item_feat.t(), None
    logit = ability @ (-item_feat[:, :A].t()) + item_feat[:, A].unsqueeze(0)
    guess = torch.sigmoid(item_feat[:, A + 1]) if irt_model == 3 else None
    return logit, guess


def _mlp3(i, h, o, last_act=None):
    layers = [nn.Linear(i, h), nn.ELU(inplace=True), nn.Linear(h, h), nn.ELU(inplace=True), nn.Linear(h, o)]
    if last_act is not None:
        layers.append(last_act)
    return nn.Sequential(*layers)


def _xavier(m):
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight.data, gain=nn.init.calculate_gain('relu'))
        nn.init.constant_(m.bias.data, 0)


class LinkedIRT(nn.Module):
    """models.py:769-813: response_mu = link(irt logit), link = 1 -> H -> H -> 1 -> sigmoid (3PL: guess mixture)."""

    def __init__(self, irt_model=1, hidden_dim=HIDDEN):
        super().__init__()
        if hidden_dim > HIDDEN:
            raise NotImplementedError(f'hidden_dim = {hidden_dim}: the per-term decoder kernel covers hidden widths up to {HIDDEN}')
        self.irt_model, self.hidden_dim = irt_model, hidden_dim
        self.link = _mlp3(1, hidden_dim, 1, nn.Sigmoid())
        self.apply(_xavier)

    def _args(self, ability, item_feat):
        logit, guess = irt_logit(self.irt_model, ability, item_feat)
        lk = self.link
        V = lk[0].bias.unsqueeze(0).expand(ability.shape[0], -1)
        return dict(U=None, V=V, W2=lk[2].weight, b2=lk[2].bias, w3=lk[4].weight.reshape(-1), b3=lk[4].bias,
                    logit=logit, w1=lk[0].weight.reshape(-1), guess=guess, resid=0.0)

    def log_lik(self, response, mask, ability, item_feat):
        return decoder_log_lik(response, mask, **self._args(ability, item_feat))

    def forward(self, ability, item_feat):
        return decoder_probs(ability.shape[0], item_feat.shape[0], **self._args(ability, item_feat)).unsqueeze(2)


class DeepIRT(nn.Module):
    """models.py:816-877: sigmoid(mlp_concat([mlp_item_feat(item), mlp_ability(ability)]))."""
    RESIDUAL = False

    def __init__(self, latent_dim, irt_model=1, hidden_dim=HIDDEN):
        super().__init__()
        if hidden_dim > HIDDEN:
            raise NotImplementedError(f'hidden_dim = {hidden_dim}: the per-term decoder kernel covers hidden widths up to {HIDDEN}')
        self.latent_dim = self.ability_dim = latent_dim
        self.irt_model, self.hidden_dim = irt_model, hidden_dim
        self.item_feat_dim = {1: 1, 2: latent_dim + 1, 3: latent_dim + 2}[irt_model]
        self.mlp_item_feat = _mlp3(self.item_feat_dim, hidden_dim, hidden_dim)
        self.mlp_ability = _mlp3(latent_dim, hidden_dim, hidden_dim)
        self.mlp_concat = _mlp3(2 * hidden_dim, hidden_dim, 1)
        self.apply(_xavier)

    def _args(self, ability, item_feat):
        mc, H = self.mlp_concat, self.hidden_dim
        # cat([hid_item, hid_ability]) through the first layer = its item half + its ability half (+ bias)
        U = self.mlp_item_feat(item_feat) @ mc[0].weight[:, :H].t()
        V = self.mlp_ability(ability) @ mc[0].weight[:, H:].t() + mc[0].bias
        a = dict(U=U, V=V, W2=mc[2].weight, b2=mc[2].bias, w3=mc[4].weight.reshape(-1), b3=mc[4].bias)
        if self.RESIDUAL:
            logit, guess = irt_logit(self.irt_model, ability, item_feat)
            a.update(logit=logit, guess=guess, resid=1.0)
        return a

    def log_lik(self, response, mask, ability, item_feat):
        return decoder_log_lik(response, mask, **self._args(ability, item_feat))

    def forward(self, ability, item_feat):
        return decoder_probs(ability.shape[0], item_feat.shape[0], **self._args(ability, item_feat)).unsqueeze(2)


class ResidualIRT(DeepIRT):
    """models.py:880-919: sigmoid(mlp_concat(...) + irt logit) (3PL: guess mixture)."""
    RESIDUAL = True

    def __init__(self, latent_dim, irt_model=1, hidden_dim=HIDDEN):
        super().__init__(latent_dim, irt_model=irt_model, hidden_dim=hidden_dim)
        self.apply(_xavier)          # (the reference's zero_init is a second xavier pass: same draws from the RNG stream)
