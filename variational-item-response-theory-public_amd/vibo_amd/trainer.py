"""Fused train step: the whole `loss = model.elbo(*model(r, m), beta); loss.backward(); adam.step()` of the
reference loop (vibo.py:243-268) as TWO kernel launches instead of ~80.

    trainer = FusedTrainer(model, lr=5e-3)
    loss = trainer.step(response, mask, beta=1.0, row_index=rows)      # device scalar, parameters updated in place

What runs (the folded step, `fold=True` with rng='native', the default of the CLI and the benchmark):
    vibo_elbo_fwd_bwd_step     the fused ELBO forward + backward over the rows (ticks Adam's step counter)
    [person-sharded: finalize inside that call, then ONE all-reduce of the flat buffer]
    vibo_train_epilogue_fused  finalize (one GPU), loss, encoder-MLP / item backward, Adam -- and the NEXT step's head: Philox
                               noise, item sample, item KL, the 2-row encoder table from the parameters just updated
The step is software-pipelined across its own iterations: when the ELBO kernel starts, everything it reads is in memory.  The
first step's head comes from vibo_fill_normal x 2 + vibo_train_prime ("priming"), repeated whenever the parameters were
changed from outside between two steps (load_state_dict, optimizers, `.copy_`: detected through the tensors' version
counters) or a larger minibatch than ever before arrives.  Writes torch does not count -- through `p.data`, through
`trainer.mlp_flat`, from another native kernel -- are NOT seen: call `trainer.invalidate()` after them (the next step then
re-primes), or build the trainer with fold=False, whose four-launch step recomputes its head from the parameters every time.
A folded forward_backward() has to be followed by update() before the next one (it raises otherwise: the second call would
tick Adam's counter and flip the double-buffered item-KL half under the pending update).  The ability-noise buffer never
moves once a hipGraph may have captured it: size it up front with `max_batch`, a larger minibatch arriving later bumps
`trainer.generation` (GraphedTrainStep re-captures when it changes).
`fold=False`, rng='torch' (noise from torch's generators: not known a step ahead) and shapes the folded step does not cover
(more than 1024 items, int64 masks, unaligned rows) take the four-launch form (vibo_train_prologue[_noise] ->
vibo_elbo_fwd_bwd = kernel + finalize -> vibo_train_epilogue); the two forms agree bit for bit (tests/test_gpu_trainer.py).
Same arithmetic as the PyTorch path (tests/test_gpu_trainer.py compares parameters after several steps); `.grad` fields are not
populated.
FusedTrainer covers the unconditional posterior without flows; `FusedTrainer(model)` returns its sibling
FusedCondFlowTrainer (same interface, vibo_ctrain_* kernels) for --conditional-posterior / --n-norm-flows models.
FusedMeanTrainer (vibo_mtrain_* kernels) is the same for --ability-merge mean with the unconditional posterior (person-sharded too).
The MLP decoders and mean x conditional train through the module + torch.optim path
(fused_trainer_covers() tells which).
"""
import ctypes

import torch

from . import _lib, ops


def fused_trainer_covers(model, hidden_dim=None):
    """True when one of the fused trainers of this module runs the model's whole train step natively: the product-of-experts
    encoder with the IRT decoder -- plain (FusedTrainer's kernels), or with the conditional posterior and / or planar flows
    (FusedCondFlowTrainer's, hidden width <= 64) -- or the --ability-merge mean encoder with the unconditional posterior
    (FusedMeanTrainer's; person-sharded too since round 5).  The MLP decoders and mean x conditional train through the module + torch.optim.Adam."""
    if getattr(model, 'generative_model', 'irt') != 'irt':
        return False
    if model.ability_merge == 'mean':          # FusedMeanTrainer: unconditional posterior, no flows, hidden width <= 128
        H = hidden_dim if hidden_dim is not None else model.ability_encoder.mlp1[0].weight.shape[0]
        return (not model.conditional_posterior and model.n_norm_flows == 0 and H <= 128
                and model.ability_dim <= _lib.MAX_ABILITY_DIM_FAST)
    if model.ability_merge != 'product':
        return False
    if model.conditional_posterior or model.n_norm_flows > 0:
        H = hidden_dim if hidden_dim is not None else model.ability_encoder.mlp[0].weight.shape[0]
        return H <= 64 and model.ability_dim <= _lib.MAX_ABILITY_DIM_FAST      # (vibo_ctrain_*: one 64-wide tile, 8 ability dims)
    return True


class FusedTrainer:
    def __new__(cls, model=None, *args, **kwargs):
        # one entry point: the conditional posterior / planar flows are served by the sibling class below
        # (model=None: copy / pickle re-create the object through cls.__new__(cls) and fill __dict__ themselves)
        if cls is FusedTrainer and model is not None and model.ability_merge == 'mean':
            return super().__new__(FusedMeanTrainer)
        if cls is FusedTrainer and model is not None and (model.conditional_posterior or model.n_norm_flows > 0):
            return super().__new__(FusedCondFlowTrainer)
        return super().__new__(cls)

    def __init__(self, model, lr=5e-3, rng='torch', seed=0, fused_noise=True, fold=True, max_batch=None):
        if model.ability_merge != 'product' or getattr(model, 'generative_model', 'irt') != 'irt':
            raise NotImplementedError('the fused trainers cover the product-of-experts encoder with the IRT decoder; '
                                      'use model.elbo_step + torch.optim.Adam otherwise')
        self.model = model
        self.fold = bool(fold)                # two launches per step (train hook + fused epilogue) where the shape allows
        self._primed_for = None               # folded step: (noise capacity, parameter versions) the next step's head was prepared for
        self._eps_cap = None                  # ... the ability-noise buffer [capacity] every step's epilogue refills
        self._eps_keep = []                   # (outgrown buffers stay alive: a captured graph may still write to them)
        self._max_batch = int(max_batch) if max_batch else 0      # persons of the largest minibatch to expect (sizes _eps_cap once)
        self.generation = 0                   # bumped when a buffer a captured hipGraph points at was replaced (re-capture then)
        self._folded_open = False             # a folded forward_backward() whose update() has not run yet
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
        self._watched = plist + [self.mlp_flat, self.item_mu, self.item_lv]
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
        self.kl_parts = torch.empty(2 * ((n_item + 63) // 64), device=dev)      # (two halves: the folded step double-buffers them)
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

    def invalidate(self):
        """Tell the folded step that the parameters were written behind torch's back (`p.data` edits, `trainer.mlp_flat`,
        another kernel): the head the previous epilogue left behind -- item sample, item KL, expert table, saved activations --
        is stale, the next step rebuilds it from the parameters as they are then (vibo_train_prime).  load_state_dict and other
        writes torch counts are detected without this call.  The pending noise draws are repeated for the same step counter.
        Also the way out after a step that died between its two halves (an exception in the all-reduce, a hipGraph capture that
        was aborted after forward_backward() had run on the host): the half-open folded step is forgotten."""
        self._primed_for = None
        self._folded_open = False
        self._pending = None

    reprime = invalidate

    @torch.no_grad()
    def step(self, response, mask, beta=None, row_index=None, eps_item=None, eps_ability=None):
        """One train step; returns the loss (device scalar).  = forward_backward(); [all-reduce]; update().
        eps_item [I, D] / eps_ability [B, A]: replay given reparameterisation noise instead of drawing it (parity tests against
        the reference's recorded steps; takes the four-launch form)."""
        raw = self.forward_backward(response, mask, beta=beta, row_index=row_index, eps_item=eps_item, eps_ability=eps_ability)
        if self.model._reducer is not None:
            self.model._reducer(raw.flat)     # person-sharded: ONE all-reduce per step
        return self.update()

    @torch.no_grad()
    def forward_backward(self, response, mask, beta=None, row_index=None, eps_item=None, eps_ability=None):
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
        ab_stream = 1 + getattr(model, '_shard_rank', 0)      # item noise: the same on every rank; ability noise: per rank
        given = eps_item is not None or eps_ability is not None
        if given and (eps_item is None or eps_ability is None):
            raise ValueError('pass both eps_item and eps_ability, or neither')
        step_bits = lib.vibo_train_step_supported(ctypes.byref(d)) if (self.fold and self.rng == 'native' and self.fused_noise and not given) else 0
        if step_bits & 1:
            return self._forward_backward_folded(d, step_bits, response, mask, code, row_index, B, ab_stream, stream)
        # ---- the four-launch form ----
        if given:
            eps_item, eps_ab = eps_item.contiguous().float(), eps_ability.contiguous().float()
            self._primed_for = None
            rc = lib.vibo_train_prologue(ctypes.byref(d), self.hidden, p(self.mlp_flat), p(self.item_mu), p(self.item_lv),
                                         p(eps_item), p(self.item_feat), p(self.table), p(self.saved_h), p(self.kl_parts),
                                         p(self._steps), stream)
            _lib.check(rc, 'vibo_train_prologue')
            raw = ops._BACKEND['elbo'](spec, response, mask, code, row_index, self.table, self.item_feat, eps_ab, None,
                                       _lib.REG_KL, True, B)
            self._pending = (d, eps_item, raw, None)
            self._folded_open = False             # (a four-launch step replaces whatever was pending)
            self.last = raw
            return raw
        # reference draw order: item eps, then ability eps (models.py:361,368)
        if self.rng == 'native':
            eps_item = self._eps_item
            # one buffer per batch size, never freed or replaced: a captured hipGraph keeps the pointer it was recorded
            # with, and the epoch's last, shorter minibatch runs eagerly in between the replays
            eps_ab = self._eps_ab.get(B)
            if eps_ab is None:
                eps_ab = self._eps_ab[B] = torch.empty(B, model.ability_dim, device=dev)
            self._primed_for = None           # (these draws move on without refilling the folded step's buffers)
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
        self._pending = (d, eps_item, raw, None)
        self._folded_open = False
        self.last = raw
        return raw

    def _param_versions(self):
        # (torch bumps a tensor's version on every in-place write it knows of -- load_state_dict, optimizers, .copy_ -- while
        #  this library's kernels update the same memory without touching it: a change means somebody else wrote)
        return tuple(t._version for t in self._watched)

    def _forward_backward_folded(self, d, step_bits, response, mask, code, row_index, B, ab_stream, stream):
        """The folded step's first launch: vibo_elbo_fwd_bwd_step (+ the stand-alone finalize when an all-reduce follows)."""
        model, spec, lib, p = self.model, self.model.spec, _lib.load(), ops._ptr
        dev = response.device
        A = model.ability_dim
        # The head of a step (noise, item sample, item KL, expert table) is left behind by the previous step's epilogue in
        # buffers that never move (a captured hipGraph keeps their pointers); the ability noise goes into ONE buffer of fixed
        # capacity -- the streams are indexed by element, so a step of fewer persons reads a prefix of the same values a fresh
        # draw would give (the epoch's last, shorter minibatch between two replays).  Before the first step, when a larger
        # batch than ever before arrives, or when somebody else wrote the parameters, the head is (re)built here.
        if self._folded_open:
            raise RuntimeError('FusedTrainer: forward_backward() was called twice without update() in between (the folded step '
                               'ticks Adam\'s counter in its first launch; use fold=False to evaluate gradients without updating)')
        need = B * A
        if self._eps_cap is None or self._eps_cap.numel() < need:
            if self._eps_cap is not None:
                # a hipGraph captured before this moment keeps reading -- and its epilogue refilling -- the old buffer, while
                # eager steps move on with the new one: captured steps have to be re-captured (GraphedTrainStep does, on
                # `generation`); pass max_batch to the constructor to never get here
                self._eps_keep.append(self._eps_cap)
                self.generation += 1
            self._eps_cap = torch.empty(max(need, self._max_batch * A), device=dev)
            self._primed_for = None
        state = (self._eps_cap.numel(),) + self._param_versions()
        if self._primed_for != state:
            noise_step = ctypes.c_void_p(self._steps.data_ptr() + 4)          # completed steps (step_count[1])
            _lib.check(lib.vibo_fill_normal(p(self._eps_item), self._eps_item.numel(), self.seed, noise_step, 0, stream), 'vibo_fill_normal')
            _lib.check(lib.vibo_fill_normal(p(self._eps_cap), self._eps_cap.numel(), self.seed, noise_step, ab_stream, stream), 'vibo_fill_normal')
            rc = lib.vibo_train_prime(ctypes.byref(d), self.hidden, p(self.mlp_flat), p(self.item_mu), p(self.item_lv), p(self._eps_item),
                                      p(self.item_feat), p(self.table), p(self.saved_h), p(self.kl_parts), p(self._steps), stream)
            _lib.check(rc, 'vibo_train_prime')
            self._primed_for = state
        eps_item, eps_ab = self._eps_item, self._eps_cap[:need].view(B, A)
        fused_finalize = bool(step_bits & 2) and model._reducer is None
        raw = ops._BACKEND['elbo'](spec, response, mask, code, row_index, self.table, self.item_feat, eps_ab, None, _lib.REG_KL, True, B,
                                   train_step=(self._steps, fused_finalize))
        self._pending = (d, eps_item, raw, ab_stream)
        self._folded_open = True
        self.last = raw
        return raw

    @property
    def step_count(self):
        """Adam's step number (device int32 scalar)."""
        return self._steps[0]

    @torch.no_grad()
    def update(self):
        """Loss, encoder-MLP / item backward and Adam from the (all-reduced) flat buffer of forward_backward()."""
        if self._pending is None:
            raise RuntimeError('FusedTrainer.update(): no forward_backward() is pending')
        d, eps_item, raw, folded_stream = self._pending
        lib, p = _lib.load(), ops._ptr
        stream = ctypes.c_void_p(torch.cuda.current_stream(raw.flat.device).cuda_stream)
        if folded_stream is not None:
            self._folded_open = False
            rc = lib.vibo_train_epilogue_fused(ctypes.byref(d), self.hidden, p(raw.workspace), p(raw.flat), p(self.saved_h),
                                               p(self.kl_parts), p(eps_item), p(self.beta), p(self.lr), p(self._steps),
                                               p(self.mlp_flat), p(self.mlp_m), p(self.mlp_v), p(self.item_mu), p(self.item_lv),
                                               p(self.item_m), p(self.item_v), p(self.loss), self.seed, p(self.item_feat),
                                               p(self.table), p(self._eps_cap), self._eps_cap.numel(), folded_stream, stream)
            _lib.check(rc, 'vibo_train_epilogue_fused')
            return self.loss
        rc = lib.vibo_train_epilogue(ctypes.byref(d), self.hidden, p(raw.flat), p(self.saved_h), p(self.kl_parts),
                                     p(eps_item), p(self.beta), p(self.lr), p(self._steps), p(self.mlp_flat),
                                     p(self.mlp_m), p(self.mlp_v), p(self.item_mu), p(self.item_lv), p(self.item_m),
                                     p(self.item_v), p(self.loss), stream)
        _lib.check(rc, 'vibo_train_epilogue')
        return self.loss


class FusedCondFlowTrainer(FusedTrainer):
    """The fused train step for --conditional-posterior and / or --n-norm-flows models (vibo.py:243-268 with
    models.py:337-354, 380-443, 664-710, flows.py:21-66): vibo_ctrain_prologue (item sample, item-side flows, ability-flow
    packing, the encoder MLP on the 2 x I rows [c, item_i] -> expert table; optionally the Philox noise) -> vibo_elbo_fwd_bwd
    -> [one all-reduce when person-sharded] -> vibo_ctrain_epilogue (loss, table-MLP / flow / sample backward, Adam on
    everything).  No PyTorch autograd node: the step replays from a hipGraph like FusedTrainer's.  Same interface
    (`FusedTrainer(model, ...)` returns this class for such models).  Hidden width <= 64."""

    def __init__(self, model, lr=5e-3, rng='torch', seed=0, fused_noise=True, fold=True, max_batch=None):
        # (fold: FusedTrainer's two-launch form; this class's step is prologue / ELBO call / epilogue either way)
        if model.ability_merge != 'product' or getattr(model, 'generative_model', 'irt') != 'irt':
            raise NotImplementedError('the fused trainers cover the product-of-experts encoder with the IRT decoder; '
                                      'use model.elbo_step + torch.optim.Adam otherwise')
        self.model = model
        self.generation = 0                   # (no buffer of this step ever moves: see FusedTrainer.generation)
        mlp = model.ability_encoder.mlp
        self.hidden = mlp[0].weight.shape[0]
        if self.hidden > 64:
            raise NotImplementedError('FusedCondFlowTrainer: hidden_dim <= 64 (one 64-wide tile of the matrix-pipe table MLP); '
                                      'use model.elbo_step + torch.optim.Adam otherwise')
        plist = [mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, mlp[4].weight, mlp[4].bias]
        F = model.n_norm_flows
        if F > 0:
            for st in (model.ability_norm_flows, model.item_norm_flows):
                for fl in st.flows:
                    plist += [fl.u, fl.w, fl.b]
        dev = plist[0].device
        # one flat buffer (the layout of include/vibo_hip.h: vibo_ctrain_*); the nn.Parameters become views of it
        self.par_flat = torch.cat([p.detach().reshape(-1) for p in plist]).contiguous()
        off = 0
        for p in plist:
            n = p.numel()
            p.data = self.par_flat[off:off + n].view_as(p)
            off += n
        self.par_m = torch.zeros_like(self.par_flat)
        self.par_v = torch.zeros_like(self.par_flat)
        self.item_mu = model.item_encoder.mu_lookup.weight
        self.item_lv = model.item_encoder.logvar_lookup.weight
        assert self.item_mu.is_contiguous() and self.item_lv.is_contiguous()
        I, D = self.item_mu.shape
        A = model.ability_dim
        n_item = I * D
        self.item_m = torch.zeros(2 * n_item, device=dev)
        self.item_v = torch.zeros(2 * n_item, device=dev)
        self._steps = torch.zeros(2, dtype=torch.int32, device=dev)
        self.lr = torch.tensor(float(lr), device=dev)
        self.beta = torch.tensor(1.0, device=dev)
        self._beta_host = 1.0
        self.item_feat = torch.empty_like(self.item_mu)
        self.item_k = torch.empty_like(self.item_mu) if F > 0 else self.item_feat
        self.table = torch.empty((2, I, 2 * A) if model.conditional_posterior else (2, 2 * A), device=dev)
        self.flow_packed = torch.empty(F, 2 * A + 1, device=dev) if F > 0 else None
        self._desc0 = ops._make_desc(model.spec, 1, I, _lib.MASK_NONE, _lib.REG_SAMPLED if F > 0 else _lib.REG_KL, True, I, 0)
        lib = _lib.load()
        if lib.vibo_ctrain_param_floats(ctypes.byref(self._desc0), self.hidden) != self.par_flat.numel():
            raise RuntimeError('FusedCondFlowTrainer: parameter layout mismatch')
        self.scratch = torch.empty(int(lib.vibo_ctrain_scratch_floats(ctypes.byref(self._desc0), self.hidden)), device=dev)
        self.loss = torch.zeros((), device=dev)
        self.last = None
        self._pending = None
        if rng not in ('torch', 'native'):
            raise ValueError("rng must be 'torch' or 'native'")
        self.rng, self.seed = rng, int(seed)
        if not fused_noise:
            raise NotImplementedError('FusedCondFlowTrainer draws the native noise inside vibo_ctrain_prologue (there is no '
                                      'separate vibo_fill_normal form of this step): fused_noise=False is not available')
        self.fused_noise = True
        self._eps_item = torch.empty_like(self.item_mu)
        self._eps_ab = {}

    @torch.no_grad()
    def forward_backward(self, response, mask, beta=None, row_index=None, eps_item=None, eps_ability=None):
        if beta is not None:
            self.set_beta(beta)
        model, spec, lib = self.model, self.model.spec, _lib.load()
        response, mask, code = ops.prepare_rows(response, mask)
        B = int(row_index.numel()) if row_index is not None else response.shape[0]
        I = response.shape[1]
        dev = response.device
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        F = model.n_norm_flows
        reg_mode = _lib.REG_SAMPLED if F > 0 else _lib.REG_KL
        d = ops._make_desc(spec, B, I, code, reg_mode, True, response.stride(0), mask.stride(0) if mask is not None else 0)
        p = ops._ptr
        ab_stream = 1 + getattr(model, '_shard_rank', 0)
        native = self.rng == 'native' and eps_item is None
        given_ab = None
        if eps_item is not None:
            if eps_ability is None:
                raise ValueError('pass both eps_item and eps_ability, or neither')
            eps_item, given_ab = eps_item.contiguous().float(), eps_ability.contiguous().float()
            eps_ab = None
        elif native:
            eps_item = self._eps_item
            eps_ab = self._eps_ab.get(B)
            if eps_ab is None:
                eps_ab = self._eps_ab[B] = torch.empty(B, model.ability_dim, device=dev)
        else:
            # reference draw order: item eps, then ability eps (models.py:361,368)
            eps_item = model._randn(self.item_mu.shape, self.item_mu, model._item_gen)
            eps_ab = None
        rc = lib.vibo_ctrain_prologue(ctypes.byref(d), self.hidden, p(self.par_flat), p(self.item_mu), p(self.item_lv),
                                      p(eps_item), self.seed, 1 if native else 0, p(eps_ab), ab_stream, p(self.item_feat),
                                      p(self.item_k), p(self.table), p(self.flow_packed), p(self.scratch), p(self._steps), stream)
        _lib.check(rc, 'vibo_ctrain_prologue')
        if given_ab is not None:
            eps_ab = given_ab
        elif not native:
            eps_ab = model._randn((B, model.ability_dim), self.item_mu, model._ability_gen)
        raw = ops._BACKEND['elbo'](spec, response, mask, code, row_index, self.table, self.item_k, eps_ab, self.flow_packed,
                                   reg_mode, True, B)
        self._pending = (d, eps_item, raw)
        self.last = raw
        return raw

    @torch.no_grad()
    def update(self):
        d, eps_item, raw = self._pending
        lib, p = _lib.load(), ops._ptr
        stream = ctypes.c_void_p(torch.cuda.current_stream(raw.flat.device).cuda_stream)
        rc = lib.vibo_ctrain_epilogue(ctypes.byref(d), self.hidden, p(raw.flat), p(eps_item), p(self.item_feat), p(self.item_k),
                                      p(self.beta), p(self.lr), p(self._steps), p(self.par_flat), p(self.par_m), p(self.par_v),
                                      p(self.item_mu), p(self.item_lv), p(self.item_m), p(self.item_v), p(self.scratch),
                                      p(self.loss), stream)
        _lib.check(rc, 'vibo_ctrain_epilogue')
        return self.loss


class FusedMeanTrainer(FusedTrainer):
    """The fused train step for --ability-merge mean models with the unconditional posterior (vibo.py:243-268 with
    models.py:584-594, 631-650): vibo_mtrain_prologue (item sample, item KL, the 2-row mlp1 forward and the u, v collapse of
    mlp2[0]; optionally the Philox noise) -> vibo_mean_encoder_forward (per-person posterior from the row counts) ->
    vibo_elbo_fwd_bwd in VIBO_POSTERIOR_GIVEN mode -> vibo_mean_encoder_backward_sets -> vibo_mtrain_epilogue (loss, the
    backward through u, v and mlp1 by hand, Adam on everything).  No PyTorch autograd node: the step replays from a hipGraph
    like FusedTrainer's (eight launches since the GIVEN call reads / writes the posterior itself: DESIGN 3.6).  Same interface (`FusedTrainer(model, ...)` returns this class for such models).
    The packed row counts of the resident matrix are computed once (ops.row_counts keeps them while the same tensors come back).
    Person-sharded (round 5): `reduce_shards()` between the two halves all-reduces [scalars | item gradient | encoder gradient sums]
    in one collective; the per-person posterior gradients stay on their rank."""

    def __init__(self, model, lr=5e-3, rng='torch', seed=0, fused_noise=True, fold=True, max_batch=None):
        if not fused_trainer_covers(model):
            raise NotImplementedError('FusedMeanTrainer: --ability-merge mean with the unconditional posterior, the IRT decoder, no '
                                      'flows, hidden_dim <= 128, ability_dim <= 8; use model.elbo_step + torch.optim.Adam otherwise')
        self.model = model
        self.generation = 0
        enc = model.ability_encoder
        self.hidden = enc.mlp1[0].weight.shape[0]
        plist = [enc.mlp1[0].weight, enc.mlp1[0].bias, enc.mlp1[2].weight, enc.mlp1[2].bias,
                 enc.mlp2[0].weight, enc.mlp2[0].bias, enc.mlp2[2].weight, enc.mlp2[2].bias]
        dev = plist[0].device
        # one flat buffer (the layout of include/vibo_hip.h: vibo_mtrain_*); the nn.Parameters become views of it
        self.par_flat = torch.cat([p.detach().reshape(-1) for p in plist]).contiguous()
        off = 0
        for p in plist:
            n = p.numel()
            p.data = self.par_flat[off:off + n].view_as(p)
            off += n
        self.par_m = torch.zeros_like(self.par_flat)
        self.par_v = torch.zeros_like(self.par_flat)
        self.item_mu = model.item_encoder.mu_lookup.weight
        self.item_lv = model.item_encoder.logvar_lookup.weight
        assert self.item_mu.is_contiguous() and self.item_lv.is_contiguous()
        I, D = self.item_mu.shape
        A, H = model.ability_dim, self.hidden
        n_item = I * D
        self._desc0 = ops._make_desc(model.spec, 1, I, _lib.MASK_NONE, _lib.REG_KL, True, I, 0)
        lib = _lib.load()
        if lib.vibo_mtrain_param_floats(ctypes.byref(self._desc0), H) != self.par_flat.numel():
            raise RuntimeError('FusedMeanTrainer: parameter layout mismatch')
        self.item_m = torch.zeros(2 * n_item, device=dev)
        self.item_v = torch.zeros(2 * n_item, device=dev)
        self._steps = torch.zeros(2, dtype=torch.int32, device=dev)
        self.lr = torch.tensor(float(lr), device=dev)
        self.beta = torch.tensor(1.0, device=dev)
        self._beta_host = 1.0
        self.item_feat = torch.empty_like(self.item_mu)
        self.uv = torch.empty(2 * H, device=dev)
        self.saved = torch.empty(4 * H, device=dev)
        self.grad_sums = torch.empty(2 * H + 2 * A * H + 2 * A, device=dev)
        self.kl_parts = torch.empty((n_item + 63) // 64, device=dev)
        o = 2 * H + H * H + H + H * H + H
        self._w22 = self.par_flat[o:o + 2 * A * H]
        self._b22 = self.par_flat[o + 2 * A * H:o + 2 * A * H + 2 * A]
        self.loss = torch.zeros((), device=dev)
        self.last = None
        self._pending = None
        if rng not in ('torch', 'native'):
            raise ValueError("rng must be 'torch' or 'native'")
        self.rng, self.seed = rng, int(seed)
        self.fused_noise = True
        self._eps_item = torch.empty_like(self.item_mu)
        self._eps_ab = {}
        self._parts = {}

    @torch.no_grad()
    def forward_backward(self, response, mask, beta=None, row_index=None, eps_item=None, eps_ability=None):
        if beta is not None:
            self.set_beta(beta)
        model, spec, lib, p = self.model, self.model.spec, _lib.load(), ops._ptr
        counts = ops.row_counts(response, mask)             # packed (n_correct << 16 | n_observed) of every resident row, cached
        if row_index is not None:
            counts = counts[row_index]
        response, mask, code = ops.prepare_rows(response, mask)
        B = int(row_index.numel()) if row_index is not None else response.shape[0]
        I = response.shape[1]
        dev = response.device
        A, H = model.ability_dim, self.hidden
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        d = ops._make_desc(spec, B, I, code, _lib.REG_KL, True, response.stride(0), mask.stride(0) if mask is not None else 0)
        given = eps_item is not None
        native = self.rng == 'native' and not given
        if given:
            if eps_ability is None:
                raise ValueError('pass both eps_item and eps_ability, or neither')
            eps_item, eps_ab = eps_item.contiguous().float(), eps_ability.contiguous().float()
        elif native:
            eps_item = self._eps_item
            eps_ab = self._eps_ab.get(B)
            if eps_ab is None:
                eps_ab = self._eps_ab[B] = torch.empty(B, A, device=dev)
        else:
            # reference draw order: item eps, then ability eps (models.py:361,368)
            eps_item = model._randn(self.item_mu.shape, self.item_mu, model._item_gen)
            eps_ab = None
        ab_stream = 1 + getattr(model, '_shard_rank', 0)      # (person-sharded: item noise identical on every rank, ability noise per rank)
        rc = lib.vibo_mtrain_prologue(ctypes.byref(d), H, p(self.par_flat), p(self.item_mu), p(self.item_lv), p(eps_item), self.seed,
                                      1 if native else 0, p(eps_ab) if native else ctypes.c_void_p(0), ab_stream, p(self.item_feat), p(self.uv),
                                      p(self.saved), p(self.kl_parts), p(self._steps), stream)
        _lib.check(rc, 'vibo_mtrain_prologue')
        if eps_ab is None:
            eps_ab = model._randn((B, A), self.item_mu, model._ability_gen)
        post = torch.empty(B, 2 * A, device=dev)
        dm = ops._mean_desc(counts, A)
        rc = lib.vibo_mean_encoder_forward(ctypes.byref(dm), H, p(counts), p(self.uv[:H]), p(self.uv[H:]), p(self._w22), p(self._b22),
                                           p(post), stream)
        _lib.check(rc, 'vibo_mean_encoder_forward')
        raw = ops._BACKEND['elbo'](spec, response, mask, code, row_index, post, self.item_feat, eps_ab, None, _lib.REG_KL, True, B)
        n_part = lib.vibo_mean_encoder_partials(ctypes.byref(dm))
        parts = self._parts.get(n_part)
        if parts is None:
            parts = self._parts[n_part] = torch.empty(n_part, 2 * H + 2 * A * H + 2 * A, device=dev)
        grad_sets = raw.flat[_lib.NUM_SCALARS:_lib.NUM_SCALARS + 2 * B * 2 * A]
        rc = lib.vibo_mean_encoder_backward_sets(ctypes.byref(dm), H, p(counts), p(self.uv[:H]), p(self.uv[H:]), p(self._w22),
                                                 p(grad_sets), p(self.beta), p(parts), n_part, stream)
        _lib.check(rc, 'vibo_mean_encoder_backward_sets')
        self._pending = (d, eps_item, raw, parts, n_part)
        self.last = raw
        return raw

    @torch.no_grad()
    def reduce_shards(self):
        """Person sharding (round 5): ONE all-reduce per step of [8 scalars | item gradient | the encoder's gradient sums] -- what
        is a sum over persons.  The per-person posterior gradients inside raw.flat (2 x B x 2A: this rank's persons) stay local;
        the per-wave encoder partial records are summed on this rank first (their count depends on the shard size)."""
        model = self.model
        if model._reducer is None:
            return
        d, eps_item, raw, parts, n_part = self._pending
        B, A = raw.ability_mu.shape[0], model.ability_dim
        ns = _lib.NUM_SCALARS
        o = ns + 2 * B * 2 * A                              # item gradient behind the per-person sets
        n_item = self.item_mu.numel()
        psum = parts[:n_part].sum(0, keepdim=True)
        buf = torch.cat([raw.flat[:ns], raw.flat[o:o + n_item], psum.reshape(-1)])
        model._reducer(buf)
        raw.flat[:ns].copy_(buf[:ns])
        raw.flat[o:o + n_item].copy_(buf[ns:ns + n_item])
        psum.copy_(buf[ns + n_item:].view_as(psum))
        self._pending = (d, eps_item, raw, psum, 1)

    @torch.no_grad()
    def step(self, response, mask, beta=None, row_index=None, eps_item=None, eps_ability=None):
        self.forward_backward(response, mask, beta=beta, row_index=row_index, eps_item=eps_item, eps_ability=eps_ability)
        self.reduce_shards()
        return self.update()

    @torch.no_grad()
    def update(self):
        d, eps_item, raw, parts, n_part = self._pending
        lib, p = _lib.load(), ops._ptr
        stream = ctypes.c_void_p(torch.cuda.current_stream(raw.flat.device).cuda_stream)
        rc = lib.vibo_mtrain_epilogue(ctypes.byref(d), self.hidden, p(raw.flat), p(parts), n_part, p(self.grad_sums), p(self.saved),
                                      p(self.kl_parts), p(eps_item), p(self.beta), p(self.lr), p(self._steps), p(self.par_flat),
                                      p(self.par_m), p(self.par_v), p(self.item_mu), p(self.item_lv), p(self.item_m), p(self.item_v),
                                      p(self.loss), stream)
        _lib.check(rc, 'vibo_mtrain_epilogue')
        return self.loss
