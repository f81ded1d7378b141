"""Drop-in VIBO_1PL / VIBO_2PL / VIBO_3PL modules whose ELBO runs in the fused HIP kernel.

Same constructor signature, ``state_dict`` keys, RNG draw order and method
surface as the reference (src/torch_core/models.py:246-548):

    model = VIBO_2PL(ability_dim, num_item, hidden_dim=64, ability_merge='product', ...)
    outputs = model(response, mask)                       # models.py:337-354
    loss = model.elbo(*outputs, annealing_factor=beta)    # models.py:380-443
    loss.backward()

``forward`` launches ONE fused kernel that already contains the backward pass;
``elbo`` only combines its two heads (log-lik, regulariser) with the item-side
terms, so ``loss.backward()`` never touches the response matrix again.

The only piece of forward()'s tuple that is not materialised is ``response_mu``
([B,I,1], as large as the input): it is a :class:`DeferredResponseMu` that
``elbo`` recognises; call ``.materialize()`` (or ``model.decode``) for the values.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, ops
from ..decoder import DeepIRT, LinkedIRT, ResidualIRT
from ..ops import ElboSpec, decode_probs, encode_posterior, fused_elbo, item_feat_dim

LOG_2PI = math.log(2.0 * math.pi)


# ---------------------------------------------------------------------------
# parameter containers (state_dict-compatible with the reference)
# ---------------------------------------------------------------------------

def _encoder_mlp(in_dim, hidden_dim, out_dim):
    return nn.Sequential(
        nn.Linear(in_dim, hidden_dim), nn.ELU(inplace=True),
        nn.Linear(hidden_dim, hidden_dim), nn.ELU(inplace=True),
        nn.Linear(hidden_dim, out_dim),
    )


class AbilityEncoder(nn.Module):
    """Holds ``mlp`` (keys ability_encoder.mlp.{0,2,4}.*; models.py:575-582).

    The network is evaluated on the 2 (unconditional) or 2 x I (conditional)
    distinct inputs a Bernoulli response can present, never per (person, item).
    """

    def __init__(self, ability_dim, item_dim, hidden_dim, conditional):
        super().__init__()
        self.ability_dim = ability_dim
        self.conditional = conditional
        if conditional:
            # the reference builds (and discards) the unconditional net first
            # (models.py:675-693); do the same so parameter init consumes the RNG identically
            _encoder_mlp(1, hidden_dim, 2 * ability_dim)
        self.mlp = _encoder_mlp(1 + (item_dim if conditional else 0), hidden_dim, 2 * ability_dim)
        # the two observed response values; a non-persistent buffer (not in state_dict) so that building the
        # table needs no host->device copy (keeps the step hipGraph-capturable)
        self.register_buffer('_response_values', torch.tensor([[0.0], [1.0]]), persistent=False)

    def expert_table(self, item_feat=None):
        """[2,2A] (row c = mlp([c])) or [2,I,2A] (entry = mlp([c, item_i]))."""
        vals = self._response_values.to(self.mlp[0].weight.dtype)
        if not self.conditional:
            return self.mlp(vals)
        I = item_feat.shape[0]
        x = torch.cat([vals.unsqueeze(1).expand(2, I, 1), item_feat.unsqueeze(0).expand(2, I, -1)], dim=2)
        return self.mlp(x.reshape(2 * I, -1)).view(2, I, -1)


class _SumGradAcrossRanks(torch.autograd.Function):
    """Identity whose backward all-reduces (sums) the gradient over the person shards: the mean-merge encoder's
    parameters see only this rank's persons through autograd."""

    @staticmethod
    def forward(ctx, x, reducer):
        ctx.reducer = reducer
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.reducer(g)
        return g, None


class MeanAbilityEncoder(nn.Module):
    """--ability-merge mean (keys ability_encoder.mlp1.{0,2}.*, mlp2.{0,2}.*; models.py:584-594, 631-650).

    mlp1 -> ELU gives one feature vector per (person, item) term, which for a Bernoulli response takes two values;
    the mean over a person's OBSERVED items is therefore (n0 h(0) + n1 h(1)) / n_obs with the counts of
    vibo_row_counts, and mlp2 -- the one dense [B,H] x [H,H] contraction of the path -- maps it to (mu, logvar)."""

    def __init__(self, ability_dim, hidden_dim, item_dim=0, conditional=False):
        super().__init__()
        self.ability_dim = ability_dim
        self.conditional = conditional

        def nets(in_dim):
            return (nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.ELU(inplace=True), nn.Linear(hidden_dim, hidden_dim)),
                    nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.ELU(inplace=True), nn.Linear(hidden_dim, 2 * ability_dim)))
        if conditional:
            nets(1)      # the reference builds (and discards) the unconditional nets first (models.py:675-693): same RNG draws
        self.mlp1, self.mlp2 = nets(1 + (item_dim if conditional else 0))
        self.register_buffer('_response_values', torch.tensor([[0.0], [1.0]]), persistent=False)

    def posterior_conditional(self, response, mask, item_feat, reducer=None, row_index=None):
        """--conditional-posterior (models.py:695-710 with _forward_mean :631-650): the per-term feature depends on the item
        too, h[c, i] = elu(mlp1([c, item_i])), so a person's mean over its observed items is the [B, 2I] x [2I, H]
        contraction of its one-hot coded row with the 2 x I feature table -- the encoder's one dense contraction, on the
        matrix pipe straight from the cell codes (ops.CodeTableSumFn: vibo_code_table_sum_forward / _backward; hidden width
        64) -- then mlp2, a plain [B, 64] GEMM.  response: fp32 rows [.., I] + mask, or CellCodes; row_index selects the minibatch."""
        I = item_feat.shape[0]
        vals = self._response_values.to(item_feat.dtype)
        x = torch.cat([vals.unsqueeze(1).expand(2, I, 1), item_feat.unsqueeze(0).expand(2, I, -1)], dim=2)
        h = F.elu(self.mlp1(x.reshape(2 * I, -1))).view(2, I, -1)                 # [2, I, H]
        l0, l2 = self.mlp2[0], self.mlp2[2]
        w0, b0, w2, b2 = l0.weight, l0.bias, l2.weight, l2.bias
        if reducer is not None:      # person-sharded: these see only this rank's persons
            h, w0, b0, w2, b2 = (_SumGradAcrossRanks.apply(t, reducer) for t in (h, w0, b0, w2, b2))
        if h.shape[2] == 64 and I <= 32767:      # (the native path's row counts are packed as n_correct << 16 | n_observed)
            hid_sum, nobs = ops._BACKEND['cond_mean_sum'](h, response, mask, row_index)
        else:
            # other hidden widths, more than 32 767 items: two dense GEMMs on the observed / correct indicator matrices
            if isinstance(response, ops.CellCodes):
                r, m = (response.rows(row_index) if row_index is not None else response).unpack()
            else:
                r = ops.prepare_response(response)
                m = None if mask is None else ops.prepare_mask(mask)[0]
                if row_index is not None:
                    r, m = r[row_index], (None if m is None else m[row_index])
            obs = torch.ones_like(r) if m is None else (m != 0).to(r.dtype)
            right = (r == 1).to(r.dtype) * obs
            hid_sum, nobs = obs @ h[0] + right @ (h[1] - h[0]), obs.sum(1)
        hid_mean = hid_sum / nobs.unsqueeze(1)             # (no observed item: 0/0 = NaN like the reference)
        return F.linear(F.elu(F.linear(hid_mean, w0, b0)), w2, b2)

    def posterior(self, counts, reducer=None):
        """[B, 2A] = (mu | logvar) of every person from the packed row counts (0 observed items -> NaN, as the
        reference's mean over an empty set, models.py:639-642)."""
        h = F.elu(self.mlp1(self._response_values.to(self.mlp1[0].weight.dtype)))        # [2, H]: the two per-term features
        l0, l2 = self.mlp2[0], self.mlp2[2]
        u = F.linear(h[0], l0.weight, l0.bias)              # first layer of mlp2 is affine in w = n_correct / n_observed
        v = F.linear(h[1] - h[0], l0.weight)
        w2, b2 = l2.weight, l2.bias
        if reducer is not None:      # person-sharded: these four see only this rank's persons
            u, v, w2, b2 = (_SumGradAcrossRanks.apply(t, reducer) for t in (u, v, w2, b2))
        return ops.MeanEncoderFn.apply(u, v, w2, b2, counts)


class ItemEncoder(nn.Module):
    """Per-item Gaussian posterior parameters (models.py:713-726)."""

    def __init__(self, num_item, item_dim):
        super().__init__()
        self.mu_lookup = nn.Embedding(num_item, item_dim)
        self.logvar_lookup = nn.Embedding(num_item, item_dim)

    def forward(self, item_index=None):
        if item_index is None:
            return self.mu_lookup.weight, self.logvar_lookup.weight
        idx = item_index.reshape(-1).long()
        return self.mu_lookup(idx), self.logvar_lookup(idx)


class PlanarFlowParams(nn.Module):
    """u, w ~ N(0,1), b = 1 (flows.py:15-19)."""

    def __init__(self, dim):
        super().__init__()
        self.u = nn.Parameter(torch.randn(dim))
        self.w = nn.Parameter(torch.randn(dim))
        self.b = nn.Parameter(torch.ones(1))

    def uhat(self):
        uw = torch.dot(self.u, self.w)
        return self.u + (F.softplus(uw) - 1.0 - uw) * self.w / torch.sum(self.w * self.w)

    def forward(self, z):
        """Planar step on rows of z (flows.py:21-41); used for the [I,D] item side."""
        uhat = self.uhat()
        t = torch.tanh(z @ self.w + self.b)
        z_new = z + uhat.unsqueeze(0) * t.unsqueeze(1)
        psi_u = (1.0 - t * t) * torch.dot(self.w, uhat)
        return z_new, torch.log(torch.abs(1.0 + psi_u) + 1e-8)


class FlowStack(nn.Module):
    """keys <name>.flows.{k}.{u,w,b} (flows.py:44-66)."""

    def __init__(self, dim, n_flows):
        super().__init__()
        self.flows = nn.ModuleList([PlanarFlowParams(dim) for _ in range(n_flows)])

    def forward(self, z):
        """(z_K, sum of log|det J|) of the rows of z: one native forward launch (and one backward launch) for the whole stack."""
        return ops._BACKEND['flow_stack'](z, self.packed())

    def packed(self):
        """[n_flows, 2*dim+1] = (uhat | w | b) rows for the kernels; uhat (flows.py:24-26) for all flows at once."""
        U = torch.stack([f.u for f in self.flows])
        W = torch.stack([f.w for f in self.flows])
        b = torch.cat([f.b for f in self.flows]).unsqueeze(1)
        uw = (U * W).sum(1, keepdim=True)
        uhat = U + (F.softplus(uw) - 1.0 - uw) * W / (W * W).sum(1, keepdim=True)
        return torch.cat([uhat, W, b], dim=1)


# ---------------------------------------------------------------------------
# deferred pieces of forward()'s tuple
# ---------------------------------------------------------------------------

class FusedContext:
    """What one fused kernel call produced, kept until elbo() consumes it."""

    def __init__(self, model, response, mask, eps_ability, table, item_in, flow_packed, reg_mode, heads):
        self.model = model
        self.response, self.mask = response, mask
        self.eps_ability = eps_ability
        self.table, self.item_in, self.flow_packed = table, item_in, flow_packed
        self.reg_mode = reg_mode
        (self.ll, self.reg, self.scalars, self.ability_mu, self.ability_logvar, self.ability,
         self.ability_k, self.ability_ladj) = heads


class DeferredResponseMu:
    """Stands in for response_mu [B,I,1] (models.py:345,351) in forward()'s tuple."""

    def __init__(self, ctx, ability, item_feat):
        self.ctx = ctx
        self._ability, self._item = ability, item_feat

    def materialize(self):
        return self.ctx.model.decode(self._ability.detach(), self._item.detach())

    def __repr__(self):
        return 'DeferredResponseMu(call .materialize() for the [B,I,1] tensor)'


class DecoderContext:
    """One forward of a --generative-model link|deep|residual model: the posterior side is ordinary autograd (it only
    needs the row counts), the log-likelihood is the decoder kernel's (vibo_amd/decoder.py), evaluated when elbo() asks."""

    def __init__(self, model, response, mask, row_index):
        self.model, self.response, self.mask, self.row_index = model, response, mask, row_index
        self._ll = None

    @property
    def ll(self):
        if self._ll is None:
            r, m = self.response, self.mask
            if self.row_index is not None:          # the decoder kernel walks dense rows: gather the minibatch here
                if isinstance(r, ops.CellCodes):
                    r = r.rows(self.row_index)
                else:
                    r = ops.prepare_response(r)[self.row_index]
                    m = None if m is None else ops.prepare_mask(m)[0][self.row_index]
            self._ll = self.model.decoder.log_lik(r, m, self.ability_k, self.item_k)
        return self._ll


# ---------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------

class VIBO_1PL(nn.Module):
    IRT = 1

    def __init__(self, latent_dim, num_item, hidden_dim=64, ability_merge='mean',
                 conditional_posterior=False, generative_model='irt', response_dist='bernoulli',
                 replace_missing_with_prior=True, n_norm_flows=0):
        super().__init__()
        if ability_merge not in ('mean', 'product'):
            raise AssertionError('ability_merge must be mean|product')   # models.py:262
        if generative_model not in ('irt', 'link', 'deep', 'residual'):
            raise AssertionError('bad generative_model')
        if response_dist not in ('bernoulli', 'gaussian'):
            raise AssertionError('bad response_dist')
        # the fused HIP path covers the 1PL/2PL/3PL logistic link with the product-of-experts
        # encoder on Bernoulli responses (BASELINE.json north_star); nothing else is in scope
        if response_dist != 'bernoulli':
            raise NotImplementedError("only --response-dist bernoulli is implemented by the HIP engine")

        self.latent_dim = self.ability_dim = latent_dim
        self.response_dim = 1
        self.hidden_dim = hidden_dim
        self.num_item = num_item
        self.ability_merge = ability_merge
        self.conditional_posterior = conditional_posterior
        self.generative_model = generative_model
        self.response_dist = response_dist
        self.replace_missing_with_prior = replace_missing_with_prior
        self.n_norm_flows = n_norm_flows
        self.irt_num = self.IRT
        self.item_feat_dim = item_feat_dim(self.IRT, latent_dim)
        self.spec = ElboSpec(irt_model=self.IRT, ability_dim=latent_dim,
                             conditional=conditional_posterior and ability_merge != 'mean',
                             drop_missing=not replace_missing_with_prior, n_flows=n_norm_flows,
                             given=ability_merge == 'mean')
        self.spec.check_supported(num_item)

        # construction order = the reference's (models.py:281-309) so seeded init matches
        if ability_merge == 'mean':
            self.ability_encoder = MeanAbilityEncoder(latent_dim, hidden_dim, self.item_feat_dim, conditional_posterior)
        else:
            self.ability_encoder = AbilityEncoder(latent_dim, self.item_feat_dim, hidden_dim, conditional_posterior)
        self.item_encoder = ItemEncoder(num_item, self.item_feat_dim)
        if n_norm_flows > 0:
            self.ability_norm_flows = FlowStack(latent_dim, n_norm_flows)
            self.item_norm_flows = FlowStack(self.item_feat_dim, n_norm_flows)
        # per-term MLP decoders (models.py:310-327, 769-919), built after the flows like the reference
        if generative_model == 'link':
            self.decoder = LinkedIRT(irt_model=self.IRT, hidden_dim=hidden_dim)
        elif generative_model == 'deep':
            self.decoder = DeepIRT(latent_dim, irt_model=self.IRT, hidden_dim=hidden_dim)
        elif generative_model == 'residual':
            self.decoder = ResidualIRT(latent_dim, irt_model=self.IRT, hidden_dim=hidden_dim)
        self.apply(self.weights_init)

        self._reducer = None          # set by enable_person_sharding()
        self._shard_world = 1
        self._last_ctx = None         # the most recent fused step (elbo() with a materialised response_mu finds it here)
        self._last_eps_item = None
        self._item_gen = None
        self._ability_gen = None

    # ---- init / RNG ---------------------------------------------------------
    @staticmethod
    def weights_init(m):
        """xavier-normal(gain=sqrt 2) on linears, zero bias (models.py:512-518)."""
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight.data, gain=nn.init.calculate_gain('relu'))
            nn.init.constant_(m.bias.data, 0)

    def enable_person_sharding(self, reducer, seed, rank, world=None):
        """Data parallelism over persons: `reducer(flat)` all-reduces (sum) the
        kernel's flat [scalars | grads] buffer in place.  Item noise must be
        identical on every rank, ability noise must differ: dedicated generators.
        MLP-decoder models (--generative-model link | deep | residual) take their gradients from autograd on this rank's
        persons: their loss counts the replicated item-side terms 1 / world times and `allreduce_grads()` (one flat
        collective per step, called by the training loop after backward) sums every parameter gradient over the ranks."""
        self._reducer = reducer
        self._shard_rank = int(rank)
        if world is None:
            import torch.distributed as dist
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self._shard_world = int(world)
        dev = self.item_encoder.mu_lookup.weight.device
        self._item_gen = torch.Generator(device=dev).manual_seed(int(seed))
        self._ability_gen = torch.Generator(device=dev).manual_seed(int(seed) + 1 + int(rank))

    def _randn(self, shape, ref, gen):
        return torch.randn(shape, dtype=ref.dtype, device=ref.device, generator=gen)

    @staticmethod
    def reparameterize_gaussian(mean, logvar, eps=None):
        """models.py:506-510."""
        std = torch.exp(0.5 * logvar)
        if eps is None:
            eps = torch.randn_like(std)
        return eps * std + mean

    # ---- pieces -------------------------------------------------------------
    def _item_side(self, eps_item=None):
        item_mu, item_lv = self.item_encoder()
        if eps_item is None:
            eps_item = self._randn(item_mu.shape, item_mu, self._item_gen)
        self._last_eps_item = eps_item          # (kept for elbo()'s re-run in the other regulariser mode)
        item_feat = eps_item * torch.exp(0.5 * item_lv) + item_mu
        return item_feat, item_mu, item_lv

    def _run_fused(self, response, mask, *, eps_item=None, eps_ability=None, reg_mode=None, row_index=None):
        """Item sample -> expert table -> fused kernel.  Draw order: item eps, then
        ability eps (models.py:361,368)."""
        item_feat, item_mu, item_lv = self._item_side(eps_item)
        if self.n_norm_flows > 0:
            item_k, item_ladj = self.item_norm_flows(item_feat)
            flow_packed = self.ability_norm_flows.packed()
        else:
            item_k, item_ladj, flow_packed = item_feat, None, None
        if self.ability_merge == 'mean':      # per-person posterior from the row counts; the kernel takes it as given
            if not isinstance(response, ops.CellCodes):
                response = ops.prepare_response(response)
                if response.shape[1] % 4 != 0 and response.stride(0) < (response.shape[1] + 3) // 4 * 4:
                    # compact ragged rows (e.g. 95 items): this path has no fallback kernel, so pad the minibatch
                    m2 = ops.prepare_mask(mask)[0]
                    if row_index is not None:
                        response, m2, row_index = response[row_index], (m2[row_index] if m2 is not None else None), None
                    response, mask = ops.pad_rows(response, m2)
            table = self._mean_posterior(response, mask, row_index, item_feat)
        else:
            table = self.ability_encoder.expert_table(item_feat if self.conditional_posterior else None)
        B = int(row_index.numel()) if row_index is not None else response.shape[0]
        if eps_ability is None:
            eps_ability = self._randn((B, self.ability_dim), item_mu, self._ability_gen)
        if reg_mode is None:
            reg_mode = _lib.REG_SAMPLED if self.n_norm_flows > 0 else _lib.REG_KL
        heads = fused_elbo(self.spec, table, item_k, flow_packed, response, mask, eps_ability,
                           reg_mode=reg_mode, row_index=row_index, reducer=self._reducer)
        ctx = FusedContext(self, response, mask, eps_ability, table, item_k, flow_packed, reg_mode, heads)
        ctx.item_feat, ctx.item_mu, ctx.item_lv = item_feat, item_mu, item_lv
        ctx.item_k, ctx.item_ladj = item_k, item_ladj
        ctx.eps_item = self._last_eps_item          # the item noise actually used, drawn here or handed in
        ctx.row_index = row_index
        self._last_ctx = ctx
        return ctx

    def _mean_posterior(self, response, mask, row_index, item_feat, counts=None, reduce_in_backward=True):
        """(mu | logvar) [B, 2A] of the --ability-merge mean encoder: from the row counts, or (conditional posterior) from
        the minibatch's rows and the item sample.  reduce_in_backward=False: the caller sums all gradients itself."""
        reducer = self._reducer if reduce_in_backward else None
        if not self.conditional_posterior:
            if counts is None:
                counts = ops.row_counts(response, mask, row_index)
            return self.ability_encoder.posterior(counts, reducer=reducer)
        return self.ability_encoder.posterior_conditional(response, mask, item_feat, reducer=reducer, row_index=row_index)

    def _posterior_from_counts(self, counts):
        """(mu, logvar) [B, A] of the unconditional product of experts (models.py:596-629, utils.py:105-113) from the packed
        row counts: every observed cell contributes one of two experts, every missing one the N(0,1) prior (or nothing)."""
        A = self.ability_dim
        table = self.ability_encoder.expert_table(None)                      # [2, 2A]: rows = response 0 / 1
        n1 = (counts >> 16).to(table.dtype).unsqueeze(1)
        nobs = (counts & 0xffff).to(table.dtype).unsqueeze(1)
        n0 = nobs - n1
        tau = 1.0 / (torch.exp(table[:, A:]) + 1e-8)                          # [2, A]
        lam = n0 * tau[0] + n1 * tau[1]
        smu = n0 * (table[0, :A] * tau[0]) + n1 * (table[1, :A] * tau[1])
        if self.replace_missing_with_prior:
            lam = lam + (self.num_item - nobs) * (1.0 / (1.0 + 1e-8))
        return smu / lam, torch.log(1.0 / lam)

    def _conditional_posterior_poe(self, response, mask, row_index, item_feat):
        """Product of experts of the conditional encoder (models.py:695-710, utils.py:105-113) for the MLP-decoder models, where
        the ability gradient does not come out of the fused ELBO kernel.  The experts' per-person sums

            [lam | s][p, :] = sum_i [cell (p, i) observed] [tau | mu tau][code_pi, i, :]      = onehot(codes) [B, 2I] x X [2I, 2A]

        are the one-hot x table contraction of csrc/vibo_cmean.hip, straight from the minibatch's cell codes on the matrix pipe
        (ops.CodeTableSumFn: vibo_code_table_sum_forward, and its transpose vibo_code_table_sum_backward for the gradient that
        reaches the 2 x I-row expert table -- and through it the encoder MLP and the item sample); X is 2 x I x 2A numbers
        of plain autograd.  The kernels' table width is 64 columns: the 2A <= 16 used ones are padded with zeros."""
        A, I = self.ability_dim, item_feat.shape[0]
        table = self.ability_encoder.expert_table(item_feat)                  # [2, I, 2A]
        tau = 1.0 / (torch.exp(table[..., A:]) + 1e-8)
        feature = torch.cat([tau, table[..., :A] * tau, table.new_zeros(2, I, 64 - 2 * A)], dim=2)      # [2, I, 64]
        sums, nobs = ops._BACKEND['cond_mean_sum'](feature, response, mask, row_index)                  # [B, 64], [B]
        lam, smu = sums[:, :A], sums[:, A:2 * A]
        if self.replace_missing_with_prior:
            lam = lam + (I - nobs.unsqueeze(1)) * (1.0 / (1.0 + 1e-8))
        return smu / lam, torch.log(1.0 / lam)

    def _run_decoder(self, response, mask, *, eps_item=None, eps_ability=None, row_index=None):
        """forward() with a per-term MLP decoder: item sample, posterior from the row counts, sample, flows."""
        # (person-sharded: plain autograd on this rank's persons; allreduce_grads() sums the gradients after backward)
        item_feat, item_mu, item_lv = self._item_side(eps_item)
        if self.ability_merge == 'mean':
            counts = None if self.conditional_posterior else ops.row_counts(response, mask, row_index)
            amu, alv = torch.chunk(self._mean_posterior(response, mask, row_index, item_feat, counts, reduce_in_backward=False), 2, dim=1)
        elif self.conditional_posterior:
            amu, alv = self._conditional_posterior_poe(response, mask, row_index, item_feat)
        else:
            amu, alv = self._posterior_from_counts(ops.row_counts(response, mask, row_index))
        if eps_ability is None:
            eps_ability = self._randn(amu.shape, item_mu, self._ability_gen)
        ctx = DecoderContext(self, response, mask, row_index)
        ctx.eps_item, ctx.eps_ability = self._last_eps_item, eps_ability
        ctx.ability_mu, ctx.ability_logvar = amu, alv
        ctx.ability = eps_ability * torch.exp(0.5 * alv) + amu
        ctx.item_feat, ctx.item_mu, ctx.item_lv = item_feat, item_mu, item_lv
        if self.n_norm_flows > 0:
            ctx.ability_k, ctx.ability_ladj = self.ability_norm_flows(ctx.ability)
            ctx.item_k, ctx.item_ladj = self.item_norm_flows(item_feat)
        else:
            ctx.ability_k, ctx.ability_ladj, ctx.item_k, ctx.item_ladj = ctx.ability, None, item_feat, None
        self._last_ctx = ctx
        return ctx

    # ---- reference method surface --------------------------------------------
    def forward(self, response, mask, eps_item=None, eps_ability=None, row_index=None):
        run = self._run_fused if self.generative_model == 'irt' else self._run_decoder
        ctx = run(response, mask, eps_item=eps_item, eps_ability=eps_ability, row_index=row_index)
        if self.n_norm_flows > 0:
            rmu = DeferredResponseMu(ctx, ctx.ability_k, ctx.item_k)
            return (response, mask, rmu, ctx.ability_k, ctx.ability, ctx.ability_mu, ctx.ability_logvar,
                    ctx.ability_ladj, ctx.item_k, ctx.item_feat, ctx.item_mu, ctx.item_lv, ctx.item_ladj)
        rmu = DeferredResponseMu(ctx, ctx.ability, ctx.item_feat)
        return (response, mask, rmu, ctx.ability, ctx.ability_mu, ctx.ability_logvar,
                ctx.item_feat, ctx.item_mu, ctx.item_lv)

    def encode(self, response, mask, row_index=None):
        """(ability, ability_mu, ability_logvar, item_feat, item_feat_mu, item_feat_logvar)
        (models.py:356-371): forward-only kernel, no gradients through the ability side."""
        item_feat, item_mu, item_lv = self._item_side()
        with torch.no_grad():
            if self.ability_merge == 'mean':
                amu, alv = torch.chunk(self._mean_posterior(response, mask, row_index, item_feat), 2, dim=1)
            else:
                table = self.ability_encoder.expert_table(item_feat if self.conditional_posterior else None)
                amu, alv = encode_posterior(self.spec, table, response, mask, row_index=row_index)
            ability = self.reparameterize_gaussian(
                amu, alv, self._randn(amu.shape, amu, self._ability_gen))
        return ability, amu, alv, item_feat, item_mu, item_lv

    def decode(self, ability, item_feat):
        """response_mu [B,I,1] (models.py:373-378, 529-533, 544-548)."""
        if self.generative_model != 'irt':
            return self.decoder(ability, item_feat)
        return decode_probs(self.spec, ability, item_feat).unsqueeze(2)

    def elbo(self, response, mask, response_mu, ability, ability_mu, ability_logvar,
             item_feat, item_feat_mu, item_feat_logvar, annealing_factor=1, use_kl_divergence=True,
             ability_k=None, item_feat_k=None, ability_logabsdetjac=None, item_logabsdetjac=None):
        """-ELBO summed over the minibatch (models.py:380-443)."""
        if isinstance(response_mu, DeferredResponseMu):
            ctx = response_mu.ctx
        else:
            # a caller that materialised response_mu through decode() first (vibo.py:379-style code): the fused step of
            # the forward() that produced `ability` is still the one to score -- the tensor itself is not needed
            last = self._last_ctx
            ctx = last if (last is not None and (ability is last.ability or ability is last.ability_k)) else None
            if ctx is None:
                raise TypeError('elbo() expects the outputs of this model\'s forward(); a response_mu tensor that does '
                                'not come from it would need a second pass over the responses')
        if isinstance(ctx, DecoderContext):
            return self._decoder_elbo(ctx, annealing_factor, use_kl_divergence)
        want_mode = _lib.REG_SAMPLED if (self.n_norm_flows > 0 or not use_kl_divergence) else _lib.REG_KL
        if want_mode != ctx.reg_mode:
            if torch.is_grad_enabled() and ctx.ll.requires_grad:
                # rare: use_kl_divergence=False asked of a KL-mode forward -> redo the step in SAMPLED mode
                ctx = self._run_fused(ctx.response, ctx.mask, eps_item=ctx.eps_item,
                                      eps_ability=ctx.eps_ability, reg_mode=want_mode, row_index=ctx.row_index)
                item_feat, item_feat_mu, item_feat_logvar = ctx.item_feat, ctx.item_mu, ctx.item_lv
                reg = ctx.reg
            else:
                sc = ctx.scalars
                reg = (sc[_lib.S_LOGQ0] - sc[_lib.S_LADJ] - sc[_lib.S_LOGP]) if want_mode == _lib.REG_SAMPLED \
                    else sc[_lib.S_KL]
        else:
            reg = ctx.reg
        ll = ctx.ll
        if self.n_norm_flows > 0:
            assert item_feat_k is not None and item_logabsdetjac is not None
            log_q_d = _normal_logpdf(item_feat, item_feat_mu, item_feat_logvar).sum() - item_logabsdetjac.sum()
            log_p_d = _std_normal_logpdf(item_feat_k).sum()
            return -(ll + log_p_d - reg - log_q_d)
        if use_kl_divergence:
            kl_d = (-0.5 * (1.0 + item_feat_logvar - item_feat_mu.pow(2) - item_feat_logvar.exp())).sum()
            return -(ll - annealing_factor * reg - annealing_factor * kl_d)
        log_q_d = _normal_logpdf(item_feat, item_feat_mu, item_feat_logvar).sum()
        log_p_d = _std_normal_logpdf(item_feat).sum()
        return -(ll + log_p_d - reg - log_q_d)

    def _decoder_elbo(self, ctx, annealing_factor, use_kl_divergence):
        """models.py:380-443 with the decoder kernel's log-likelihood; the small per-person / per-item terms are autograd.
        Person-sharded: this rank's persons and 1 / world of the (replicated) item-side terms -- the ranks' losses and
        gradients add up to the unsharded ones (allreduce_grads)."""
        ll = ctx.ll
        iw = 1.0 / self._shard_world if self._reducer is not None else 1.0
        if self.n_norm_flows > 0:
            log_q = (_normal_logpdf(ctx.ability, ctx.ability_mu, ctx.ability_logvar).sum() - ctx.ability_ladj.sum()
                     + iw * (_normal_logpdf(ctx.item_feat, ctx.item_mu, ctx.item_lv).sum() - ctx.item_ladj.sum()))
            log_p = ll + _std_normal_logpdf(ctx.ability_k).sum() + iw * _std_normal_logpdf(ctx.item_k).sum()
            return -(log_p - log_q)
        if use_kl_divergence:
            kl_u = (-0.5 * (1.0 + ctx.ability_logvar - ctx.ability_mu.pow(2) - ctx.ability_logvar.exp())).sum()
            kl_d = (-0.5 * (1.0 + ctx.item_lv - ctx.item_mu.pow(2) - ctx.item_lv.exp())).sum()
            return -(ll - annealing_factor * kl_u - annealing_factor * iw * kl_d)
        log_p = ll + _std_normal_logpdf(ctx.ability).sum() + iw * _std_normal_logpdf(ctx.item_feat).sum()
        log_q = (_normal_logpdf(ctx.ability, ctx.ability_mu, ctx.ability_logvar).sum()
                 + iw * _normal_logpdf(ctx.item_feat, ctx.item_mu, ctx.item_lv).sum())
        return -(log_p - log_q)

    @property
    def needs_grad_allreduce(self):
        """True for a person-sharded MLP-decoder model: call allreduce_grads() between backward() and the optimizer step."""
        return self._reducer is not None and self.generative_model != 'irt'

    def allreduce_grads(self, loss=None):
        """Sum every parameter gradient (and, if given, the detached loss) over the person shards: ONE flat collective."""
        ps = [p for p in self.parameters() if p.requires_grad]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps]
                         + ([loss.detach().reshape(1)] if loss is not None else []))
        self._reducer(flat)
        off = 0
        for p in ps:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        return flat[off] if loss is not None else None

    def allreduce_loss(self, loss):
        """The detached shard-local loss of a person-sharded MLP-decoder model summed over the ranks (evaluation loops,
        vibo.py:291-312 under sharding: one scalar collective per step)."""
        t = loss.detach().reshape(1).clone()
        self._reducer(t)
        return t[0]

    def log_marginal(self, response, mask, num_samples=100, eps_item=None, eps_ability=None):
        """Importance-weighted bound with batch-level weights (models.py:445-504).  eps_item [S,I,D] / eps_ability
        [S,B,A] replay a fixed noise sequence (tests); by default it is drawn item-then-ability per sample like the
        reference's loop.  Unconditional posterior: the S forwards share one pass over the responses
        (vibo_elbo_multi_forward); otherwise one forward launch per sample."""
        with torch.no_grad():
            S = int(num_samples)
            if self.generative_model != 'irt':
                log_w = []
                for s in range(S):
                    ctx = self._run_decoder(response, mask, eps_item=None if eps_item is None else eps_item[s],
                                            eps_ability=None if eps_ability is None else eps_ability[s])
                    log_w.append(-self._decoder_elbo(ctx, 1.0, False))
                return torch.logsumexp(torch.stack(log_w), 0) - math.log(S)
            if not self.conditional_posterior and self.ability_merge == 'product':
                if not isinstance(response, ops.CellCodes):
                    response = ops.prepare_response(response)
                    if response.shape[1] % 4 != 0 and response.stride(0) < (response.shape[1] + 3) // 4 * 4:
                        # (a compact copy of e.g. 95-item rows: pad the minibatch so the multi-sample kernel applies)
                        response, mask = ops.pad_rows(response, ops.prepare_mask(mask)[0])
                log_w, eps_item, eps_ability = self._log_weights_multi(response, mask, S, eps_item, eps_ability)
                if log_w is not None:
                    return torch.logsumexp(log_w, 0) - math.log(S)
                # not covered (fewer than 4 items, ...): the loop below replays the noise already drawn
            log_w = []
            for s in range(S):
                ctx = self._run_fused(response, mask, reg_mode=_lib.REG_SAMPLED,
                                      eps_item=None if eps_item is None else eps_item[s],
                                      eps_ability=None if eps_ability is None else eps_ability[s])
                log_q_d = _normal_logpdf(ctx.item_feat, ctx.item_mu, ctx.item_lv).sum()
                if ctx.item_ladj is not None:
                    log_q_d = log_q_d - ctx.item_ladj.sum()
                log_p_d = _std_normal_logpdf(ctx.item_k).sum()
                log_w.append(ctx.ll + log_p_d - ctx.reg - log_q_d)
            log_w = torch.stack(log_w)
            return torch.logsumexp(log_w, 0) - math.log(S)

    def _log_weights_multi(self, response, mask, S, eps_item, eps_ability):
        """log w_s for s < S through the multi-sample forward kernel, or None if the configuration is not covered."""
        response, mask2, code = ops.prepare_rows(response, mask)
        if code == _lib.MASK_I64:
            return None, eps_item, eps_ability
        B = response.shape[0]
        item_mu, item_lv = self.item_encoder()
        items, log_qd, log_pd, eps_ab, drawn_items = [], [], [], [], []
        for s in range(S):                      # draw order of the reference's loop: item eps, then ability eps
            e_i = self._randn(item_mu.shape, item_mu, self._item_gen) if eps_item is None else eps_item[s]
            drawn_items.append(e_i)
            item_feat = e_i * torch.exp(0.5 * item_lv) + item_mu
            lq = _normal_logpdf(item_feat, item_mu, item_lv).sum()
            item_k = item_feat
            if self.n_norm_flows > 0:
                item_k, item_ladj = self.item_norm_flows(item_feat)
                lq = lq - item_ladj.sum()
            items.append(item_k)
            log_qd.append(lq)
            log_pd.append(_std_normal_logpdf(item_k).sum())
            eps_ab.append(self._randn((B, self.ability_dim), item_mu, self._ability_gen)
                          if eps_ability is None else eps_ability[s])
        table = self.ability_encoder.expert_table(None)
        flow_packed = self.ability_norm_flows.packed() if self.n_norm_flows > 0 else None
        sc = ops._BACKEND['multi'](self.spec, response, mask2, code, None, table.contiguous(),
                                   torch.stack(items).contiguous(), torch.stack(eps_ab).contiguous(), flow_packed,
                                   _lib.REG_SAMPLED, B)
        if sc is None:
            return None, (torch.stack(drawn_items) if eps_item is None else eps_item), torch.stack(eps_ab)
        return sc[:, _lib.S_LL] - sc[:, _lib.S_REG] + torch.stack(log_pd) - torch.stack(log_qd), eps_item, eps_ability

    # ---- fast path for training loops (no tuple round trip) -------------------
    def elbo_step(self, response, mask, annealing_factor=1.0, row_index=None):
        """loss = model.elbo(*model(response, mask), annealing_factor) in one call."""
        out = self.forward(response, mask, row_index=row_index)
        if self.n_norm_flows > 0:
            (r, m, rmu, ak, a0, amu, alv, aladj, ik, i0, imu, ilv, iladj) = out
            loss = self.elbo(r, m, rmu, a0, amu, alv, i0, imu, ilv, annealing_factor=annealing_factor,
                             use_kl_divergence=False, ability_k=ak, item_feat_k=ik,
                             ability_logabsdetjac=aladj, item_logabsdetjac=iladj)
        else:
            loss = self.elbo(*out, annealing_factor=annealing_factor, use_kl_divergence=True)
        # the step's context (autograd graph, row references, the decoder kernel's gradient records) has been consumed: do not
        # keep it alive until the next forward (it exists for elbo() calls with a materialised response_mu)
        self._last_ctx = None
        return loss


class VIBO_2PL(VIBO_1PL):
    IRT = 2


class VIBO_3PL(VIBO_2PL):
    IRT = 3


# ---------------------------------------------------------------------------
# un-amortized VI (reference models.py:100-243; training script vi.py): per-person posteriors in two embeddings
# ---------------------------------------------------------------------------

class VI_1PL(nn.Module):
    """Drop-in for the reference's VI_1PL/2PL/3PL: forward(index, response, mask) -> the 9-tuple, elbo(*outputs).
    The per-person (mu, logvar) rows looked up by `index` go to the fused kernel as a caller-supplied posterior
    (VIBO_POSTERIOR_GIVEN); its per-person gradients flow back into the embeddings through autograd."""
    IRT = 1

    def __init__(self, latent_dim, num_person, num_item):
        super().__init__()
        self.latent_dim = self.ability_dim = latent_dim
        self.response_dim = 1
        self.num_person, self.num_item = num_person, num_item
        self.item_feat_dim = item_feat_dim(self.IRT, latent_dim)
        self.spec = ElboSpec(irt_model=self.IRT, ability_dim=latent_dim, given=True)
        self.spec.check_supported(num_item)
        # construction order = the reference's (models.py:113-117): N(0,1) embeddings drawn in this order
        self.ability_mu_lookup = nn.Embedding(num_person, latent_dim)
        self.ability_logvar_lookup = nn.Embedding(num_person, latent_dim)
        self.item_mu_lookup = nn.Embedding(num_item, self.item_feat_dim)
        self.item_logvar_lookup = nn.Embedding(num_item, self.item_feat_dim)

    @staticmethod
    def reparameterize_gaussian(mean, logvar, eps=None):
        std = torch.exp(0.5 * logvar)
        return (torch.randn_like(std) if eps is None else eps) * std + mean

    def _posterior_rows(self, index):
        idx = index.reshape(-1).long()
        return self.ability_mu_lookup(idx), self.ability_logvar_lookup(idx)

    def _run(self, index, response, mask, eps_item=None, eps_ability=None, reg_mode=_lib.REG_KL, row_index=None):
        item_mu, item_lv = self.item_mu_lookup.weight, self.item_logvar_lookup.weight
        item_feat = self.reparameterize_gaussian(item_mu, item_lv, eps_item)          # item eps first (models.py:128-135)
        amu, alv = self._posterior_rows(index)
        B = amu.shape[0]
        if eps_ability is None:
            eps_ability = torch.randn(B, self.ability_dim, dtype=amu.dtype, device=amu.device)
        if not isinstance(response, ops.CellCodes):
            response = ops.prepare_response(response)
            if response.shape[1] % 4 != 0 and response.stride(0) < (response.shape[1] + 3) // 4 * 4:
                m2 = ops.prepare_mask(mask)[0]           # compact ragged rows: this mode has no fallback kernel
                if row_index is not None:
                    response, m2, row_index = response[row_index], (m2[row_index] if m2 is not None else None), None
                response, mask = ops.pad_rows(response, m2)
        table = torch.cat([amu, alv], dim=1)
        heads = fused_elbo(self.spec, table, item_feat, None, response, mask, eps_ability, reg_mode=reg_mode,
                           row_index=row_index)
        ctx = FusedContext(self, response, mask, eps_ability, table, item_feat, None, reg_mode, heads)
        ctx.item_feat, ctx.item_mu, ctx.item_lv, ctx.eps_item = item_feat, item_mu, item_lv, eps_item
        ctx.index, ctx.row_index, ctx.amu, ctx.alv = index, row_index, amu, alv
        return ctx

    def forward(self, index, response, mask, eps_item=None, eps_ability=None, row_index=None):
        """(response, mask, response_mu, ability, ability_mu, ability_logvar, item_feat, item_feat_mu,
        item_feat_logvar) (models.py:120-126).  `row_index` gathers the rows from a resident matrix in-kernel."""
        ctx = self._run(index, response, mask, eps_item, eps_ability, row_index=row_index)
        return (response, mask, DeferredResponseMu(ctx, ctx.ability, ctx.item_feat), ctx.ability, ctx.amu, ctx.alv,
                ctx.item_feat, ctx.item_mu, ctx.item_lv)

    def encode(self, index, response=None, mask=None):
        """models.py:127-139 (plain lookups + reparameterisation; no pass over the responses)."""
        item_mu, item_lv = self.item_mu_lookup.weight, self.item_logvar_lookup.weight
        item_feat = self.reparameterize_gaussian(item_mu, item_lv)
        amu, alv = self._posterior_rows(index)
        return self.reparameterize_gaussian(amu, alv), amu, alv, item_feat, item_mu, item_lv

    def decode(self, ability, item_feat):
        return decode_probs(self.spec, ability, item_feat).unsqueeze(2)

    def elbo(self, response, mask, response_mu, ability, ability_mu, ability_logvar, item_feat, item_feat_mu,
             item_feat_logvar, annealing_factor=1, use_kl_divergence=True):
        """-ELBO summed over the minibatch (models.py:144-172)."""
        if not isinstance(response_mu, DeferredResponseMu):
            raise TypeError('elbo() expects the outputs of this model\'s forward()')
        ctx = response_mu.ctx
        want_mode = _lib.REG_KL if use_kl_divergence else _lib.REG_SAMPLED
        if want_mode != ctx.reg_mode:
            if torch.is_grad_enabled() and ctx.ll.requires_grad:
                ctx = self._run(ctx.index, ctx.response, ctx.mask, ctx.eps_item, ctx.eps_ability, reg_mode=want_mode,
                                row_index=ctx.row_index)
                item_feat, item_feat_mu, item_feat_logvar = ctx.item_feat, ctx.item_mu, ctx.item_lv
                reg = ctx.reg
            else:
                sc = ctx.scalars
                reg = (sc[_lib.S_LOGQ0] - sc[_lib.S_LOGP]) if want_mode == _lib.REG_SAMPLED else sc[_lib.S_KL]
        else:
            reg = ctx.reg
        if use_kl_divergence:
            kl_d = (-0.5 * (1.0 + item_feat_logvar - item_feat_mu.pow(2) - item_feat_logvar.exp())).sum()
            return -(ctx.ll - annealing_factor * reg - annealing_factor * kl_d)
        log_q_d = _normal_logpdf(item_feat, item_feat_mu, item_feat_logvar).sum()
        log_p_d = _std_normal_logpdf(item_feat).sum()
        return -(ctx.ll + log_p_d - reg - log_q_d)

    def log_marginal(self, index, response, mask, num_samples=100):
        """Batch-level importance-weighted bound (models.py:174-207; the reference's own loop there omits `index` in its
        forward call and cannot run -- this is the same estimator with the index passed)."""
        with torch.no_grad():
            log_w = []
            for _ in range(int(num_samples)):
                outs = self.forward(index, response, mask)
                log_w.append(-self.elbo(*outs, annealing_factor=1, use_kl_divergence=False))
            return torch.logsumexp(torch.stack(log_w), 0) - math.log(int(num_samples))


class VI_2PL(VI_1PL):
    IRT = 2


class VI_3PL(VI_2PL):
    IRT = 3


# ---------------------------------------------------------------------------
# maximum-likelihood point estimates (reference models.py:22-97; training script mle.py)
# ---------------------------------------------------------------------------

class MLE_1PL(nn.Module):
    """Drop-in for the reference's MLE_1PL/2PL/3PL (keys ability.weight [P,A], item_feat.weight [I,D]).

    forward(index, response, mask) returns the materialised response_mu [B,I,1] with autograd, as the reference's
    training loop needs it (mle.py:193-197 computes the masked BCE itself) -- that is an O(B I) PyTorch tensor by
    contract.  `nll_step` is the same loss (mean over ALL B x I cells of mask x BCE) through the fused kernel: the persons'
    rows are a caller-supplied posterior with zero noise, so the sample is the point estimate and the kernel's d LL/d mu,
    d LL/d item are the gradients; one pass over the response rows, nothing of size B x I is stored."""
    IRT = 1

    def __init__(self, latent_dim, num_person, num_item):
        super().__init__()
        self.latent_dim = self.ability_dim = latent_dim
        self.response_dim = 1
        self.num_person, self.num_item = num_person, num_item
        self.item_feat_dim = item_feat_dim(self.IRT, latent_dim)
        self.spec = ElboSpec(irt_model=self.IRT, ability_dim=latent_dim, given=True)
        self.spec.check_supported(num_item)
        self.ability = nn.Embedding(num_person, latent_dim)                 # models.py:41-42, N(0,1) init
        self.item_feat = nn.Embedding(num_item, self.item_feat_dim)

    def encode(self, index, response=None, mask=None):
        return self.ability(index.reshape(-1).long()), self.item_feat.weight       # models.py:49-53

    def decode(self, ability, item_feat):
        return decode_probs(self.spec, ability, item_feat).unsqueeze(2)

    def forward(self, index, response=None, mask=None):
        ability, item = self.encode(index)
        A = self.ability_dim
        if self.IRT == 1:                                                       # models.py:729-735
            logit = ability.sum(1, keepdim=True) + item[:, 0].unsqueeze(0)
        else:                                                                    # models.py:738-766
            logit = -(ability @ item[:, :A].t()) + item[:, A].unsqueeze(0)
        p = torch.sigmoid(logit)
        if self.IRT == 3:
            guess = torch.sigmoid(item[:, A + 1]).unsqueeze(0)
            p = guess + (1.0 - guess) * p
        return p.unsqueeze(2)

    def nll_step(self, index, response, mask, row_index=None):
        """mean_{B x I}(mask * BCE(response_mu, response)) (mle.py:193-196) without materialising response_mu."""
        ability, item = self.encode(index)
        B = ability.shape[0]
        if not isinstance(response, ops.CellCodes):
            response = ops.prepare_response(response)
            if response.shape[1] % 4 != 0 and response.stride(0) < (response.shape[1] + 3) // 4 * 4:
                m2 = ops.prepare_mask(mask)[0]
                if row_index is not None:
                    response, m2, row_index = response[row_index], (m2[row_index] if m2 is not None else None), None
                response, mask = ops.pad_rows(response, m2)
        table = torch.cat([ability, torch.zeros_like(ability)], dim=1)          # (mu | logvar = 0), eps = 0: theta = mu
        heads = fused_elbo(self.spec, table, item, None, response, mask, torch.zeros_like(ability), reg_mode=_lib.REG_KL,
                           row_index=row_index)
        return -heads[0] / float(B * self.num_item)


class MLE_2PL(MLE_1PL):
    IRT = 2


class MLE_3PL(MLE_2PL):
    IRT = 3


def _normal_logpdf(x, mu, logvar):
    return -0.5 * LOG_2PI - 0.5 * logvar - 0.5 * (x - mu) ** 2 / logvar.exp()


def _std_normal_logpdf(x):
    return -0.5 * LOG_2PI - 0.5 * x * x
