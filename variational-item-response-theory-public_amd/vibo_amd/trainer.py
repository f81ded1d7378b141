"""Fused train step: the whole `loss = model.elbo(*model(r, m), beta); loss.backward(); adam.step()` of the
reference loop (vibo.py:243-268) as ~7 kernel launches instead of ~80.

    trainer = FusedTrainer(model, lr=5e-3)
    loss = trainer.step(response, mask, beta=1.0, row_index=rows)      # device scalar, parameters updated in place

What runs: torch.randn or vibo_fill_normal (item eps) -> vibo_train_prologue (item sample, item KL, encoder table) -> torch.randn
(ability eps) -> vibo_elbo_fwd_bwd (fused ELBO forward+backward) -> [one all-reduce when person-sharded] ->
vibo_train_epilogue (loss, encoder-MLP backward, item backward, Adam).  Same arithmetic as the PyTorch path
(tests/test_gpu_trainer.py compares parameters after several steps); `.grad` fields are not populated.
Applies to the unconditional posterior without flows; other configurations use the module + torch.optim path.
"""
import ctypes

import torch

from . import _lib, ops


class FusedTrainer:
    def __init__(self, model, lr=5e-3, rng='torch', seed=0, fused_noise=True):
        if model.conditional_posterior or model.n_norm_flows > 0 or model.ability_merge != 'product':
            raise NotImplementedError('FusedTrainer covers the unconditional product-of-experts posterior without flows; '
                                      'use model.elbo_step + torch.optim.Adam otherwise')
        self.model = model
        mlp = model.ability_encoder.mlp
        self.hidden = mlp[0].weight.shape[0]
        if self.hidden > 256:
            raise NotImplementedError('hidden_dim > 256')
        plist = [mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, mlp[4].weight, mlp[4].bias]
        dev = plist[0].device
        # one flat buffer  W0 | b0 | W1 | b1 | W2 | b2 ; the nn.Parameters become views of it (state_dict unchanged)
        self.mlp_flat = torch.cat([p.detach().reshape(-1) for p in plist]).contiguous()
        off = 0
        for p in plist:
            n = p.numel()
            p.data = self.mlp_flat[off:off + n].view_as(p)
            off += n
        self.mlp_m = torch.zeros_like(self.mlp_flat)
        self.mlp_v = torch.zeros_like(self.mlp_flat)
        self.item_mu = model.item_encoder.mu_lookup.weight
        self.item_lv = model.item_encoder.logvar_lookup.weight
        assert self.item_mu.is_contiguous() and self.item_lv.is_contiguous()
        n_item = self.item_mu.numel()
        self.item_m = torch.zeros(2 * n_item, device=dev)
        self.item_v = torch.zeros(2 * n_item, device=dev)
        self._steps = torch.zeros(2, dtype=torch.int32, device=dev)      # [Adam step t, completed steps (noise counter)]
        self.lr = torch.tensor(float(lr), device=dev)
        self.beta = torch.tensor(1.0, device=dev)
        self._beta_host = 1.0
        A = model.ability_dim
        self.item_feat = torch.empty_like(self.item_mu)
        self.table = torch.empty(2, 2 * A, device=dev)
        self.saved_h = torch.empty(4 * self.hidden, device=dev)
        self.kl_parts = torch.empty((n_item + 255) // 256, device=dev)
        self.loss = torch.zeros((), device=dev)
        self.last = None                      # RawElbo of the last step (posterior outputs, scalars)
        self._pending = None
        # (person-sharded: item noise is the same on every rank, ability noise uses stream 1 + rank)
        # reparameterisation noise: 'torch' = torch.randn on the model's generators (the reference's stream for a
        # given seed), 'native' = vibo_fill_normal (Philox4x32-10 keyed by `seed`, ~5x faster on [1M, 8])
        if rng not in ('torch', 'native'):
            raise ValueError("rng must be 'torch' or 'native'")
        self.rng, self.seed = rng, int(seed)
        self.fused_noise = bool(fused_noise)      # rng='native': draw the noise inside the prologue launch (2 launches fewer)
        self._eps_item = torch.empty_like(self.item_mu) if rng == 'native' else None
        self._eps_ab = {}

    def set_beta(self, beta):
        """KL weight (vibo.py:223-230).  A device scalar: update it between graph replays when annealing."""
        if float(beta) != self._beta_host:
            self.beta.fill_(float(beta))
            self._beta_host = float(beta)

    @torch.no_grad()
    def step(self, response, mask, beta=None, row_index=None):
        """One train step; returns the loss (device scalar).  = forward_backward(); [all-reduce]; update()."""
        raw = self.forward_backward(response, mask, beta=beta, row_index=row_index)
        if self.model._reducer is not None:
            self.model._reducer(raw.flat)     # person-sharded: ONE all-reduce per step
        return self.update()

    @torch.no_grad()
    def forward_backward(self, response, mask, beta=None, row_index=None):
        """Noise, prologue and the fused ELBO forward+backward of this rank's persons.  Returns the RawElbo whose
        `.flat` buffer [scalars | grads] a person-sharded caller all-reduces before `update()`.  (Split from
        `update()` so that a multi-GPU loop can replay the two halves as hipGraphs around an eager collective.)"""
        if beta is not None:
            self.set_beta(beta)
        model, spec, lib = self.model, self.model.spec, _lib.load()
        response, mask, code = ops.prepare_rows(response, mask)
        B = int(row_index.numel()) if row_index is not None else response.shape[0]
        I = response.shape[1]
        dev = response.device
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        d = ops._make_desc(spec, B, I, code, _lib.REG_KL, True, response.stride(0), mask.stride(0) if mask is not None else 0)
        p = ops._ptr
        # reference draw order: item eps, then ability eps (models.py:361,368)
        ab_stream = 1 + getattr(model, '_shard_rank', 0)      # item noise: the same on every rank; ability noise: per rank
        if self.rng == 'native':
            eps_item = self._eps_item
            # one buffer per batch size, never freed or replaced: a captured hipGraph keeps the pointer it was recorded
            # with, and the epoch's last, shorter minibatch runs eagerly in between the replays
            eps_ab = self._eps_ab.get(B)
            if eps_ab is None:
                eps_ab = self._eps_ab[B] = torch.empty(B, model.ability_dim, device=dev)
        else:
            eps_item = model._randn(self.item_mu.shape, self.item_mu, model._item_gen)
        if self.rng == 'native' and self.fused_noise:       # noise drawn inside the prologue launch
            rc = lib.vibo_train_prologue_noise(ctypes.byref(d), self.hidden, p(self.mlp_flat), p(self.item_mu), p(self.item_lv),
                                               p(eps_item), p(self.item_feat), p(self.table), p(self.saved_h), p(self.kl_parts),
                                               p(self._steps), self.seed, p(eps_ab), ab_stream, stream)
            _lib.check(rc, 'vibo_train_prologue_noise')
        else:
            noise_step = ctypes.c_void_p(self._steps.data_ptr() + 4)          # completed steps (step_count[1])
            if self.rng == 'native':
                _lib.check(lib.vibo_fill_normal(p(eps_item), eps_item.numel(), self.seed, noise_step, 0, stream), 'vibo_fill_normal')
                _lib.check(lib.vibo_fill_normal(p(eps_ab), eps_ab.numel(), self.seed, noise_step, ab_stream, stream), 'vibo_fill_normal')
            rc = lib.vibo_train_prologue(ctypes.byref(d), self.hidden, p(self.mlp_flat), p(self.item_mu), p(self.item_lv),
                                         p(eps_item), p(self.item_feat), p(self.table), p(self.saved_h), p(self.kl_parts),
                                         p(self._steps), stream)
            _lib.check(rc, 'vibo_train_prologue')
            if self.rng != 'native':
                eps_ab = model._randn((B, model.ability_dim), self.item_mu, model._ability_gen)
        raw = ops._BACKEND['elbo'](spec, response, mask, code, row_index, self.table, self.item_feat, eps_ab, None,
                                   _lib.REG_KL, True, B)
        self._pending = (d, eps_item, raw)
        self.last = raw
        return raw

    @property
    def step_count(self):
        """Adam's step number (device int32 scalar)."""
        return self._steps[0]

    @torch.no_grad()
    def update(self):
        """Loss, encoder-MLP / item backward and Adam from the (all-reduced) flat buffer of forward_backward()."""
        d, eps_item, raw = self._pending
        lib, p = _lib.load(), ops._ptr
        stream = ctypes.c_void_p(torch.cuda.current_stream(raw.flat.device).cuda_stream)
        rc = lib.vibo_train_epilogue(ctypes.byref(d), self.hidden, p(raw.flat), p(self.saved_h), p(self.kl_parts),
                                     p(eps_item), p(self.beta), p(self.lr), p(self._steps), p(self.mlp_flat),
                                     p(self.mlp_m), p(self.mlp_v), p(self.item_mu), p(self.item_lv), p(self.item_m),
                                     p(self.item_v), p(self.loss), stream)
        _lib.check(rc, 'vibo_train_epilogue')
        return self.loss
