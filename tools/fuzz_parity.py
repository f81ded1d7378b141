#!/usr/bin/env python3
"""Randomised parity sweep: fused kernels (through the C ABI) vs the CPU analytic oracle on random configurations.
   python tools/fuzz_parity.py [--seconds 120] [--seed 0]      (needs the MI355X; test infrastructure, not product)"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'variational-item-response-theory-public_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from oracle import vibo_oracle as O
from oracle import vibo_table_ref as T
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=120)
ap.add_argument('--seed', type=int, default=0)
ap.add_argument('--target', choices=['elbo', 'multi', 'module', 'trainer'], default='elbo', help="'multi': vibo_elbo_multi_forward vs one forward launch per sample")
ap.add_argument('--focus', choices=['', 'cond1'], default='', help="'cond1': conditional posterior at ability_dim 1 on fp32 rows under the matrix-kernel pin -- the mode with the first pass folded into the matrix kernel (its XM == 3)")
ap.add_argument('--replay', type=str, default='', help='"irt A B I cond flows drop missing pad scale dataseed gather no_mask fwd_only codes given kflag" of a reported failure')
a = ap.parse_args()
rng = random.Random(a.seed)
d = torch.device('cuda:0')


def rel(x, y):
    x, y = torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(y, dtype=torch.float64)
    return float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))


def fuzz_multi():
    t0, n, worst = time.time(), 0, 0.0
    while time.time() - t0 < a.seconds:
        irt = rng.choice([1, 2, 2, 3])
        A = rng.choice([1, 2, 3, 4, 5, 8])
        I = rng.choice([4, 64, 100, 256, 260, 512, 1000, 1024, 1028, 2500])
        B = rng.choice([1, 8, 9, 17, 64, 130])
        S = rng.choice([1, 2, 3, 4, 5, 7, 9])
        n_flows = rng.choice([0, 0, 2, 4])
        missing = rng.choice([0.0, 0.2])
        spec = ElboSpec(irt_model=irt, ability_dim=A, n_flows=n_flows)
        g = torch.Generator().manual_seed(rng.randrange(1 << 30))
        resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=missing)
        table = torch.randn(2, 2 * A, generator=g) * 0.7
        items = torch.randn(S, I, spec.item_dim, generator=g) * 0.6
        eps = torch.randn(S, B, A, generator=g)
        fl = None
        if n_flows:           # uhat | w | b with the reference's constraint w.uhat >= -1 (flows.py:23-26): without it 1 + psi.u crosses
            raw = torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5          # zero and log|.| amplifies fp32 rounding without bound
            u, w = raw[:, :A], raw[:, A:2 * A]
            wu = (w * u).sum(1, keepdim=True)
            raw[:, :A] = u + (-1 + torch.nn.functional.softplus(wu) - wu) * w / (w * w).sum(1, keepdim=True)
            fl = raw.to(d)
        r_, m_ = ops.pad_rows(resp.to(d), mask.bool().to(d))
        r = ops.prepare_response(r_)
        m, code = ops.prepare_mask(m_)
        sc = ops._hip_multi_forward(spec, r, m, code, None, table.to(d), items.to(d), eps.to(d), fl, _lib.REG_SAMPLED, B)
        assert sc is not None
        for s_ in range(S):
            one = ops._hip_launch_elbo(spec, r, m, code, None, table.to(d), items[s_].to(d).contiguous(), eps[s_].to(d).contiguous(),
                                       fl, _lib.REG_SAMPLED, False, B)
            x, y = sc[s_, :7].cpu().double(), one.scalars[:7].cpu().double()
            e = float((x - y).abs().max()) / max(1.0, float(y.abs().max()))
            worst = max(worst, e)
            if not e < 3e-6:
                print(f'FAIL multi irt={irt} A={A} B={B} I={I} S={S} flows={n_flows} missing={missing} sample={s_}: {e} {x} {y}')
                sys.exit(1)
        n += 1
    print(f'fuzz multi ok: {n} random configurations, worst relative error {worst:.2e}')


def fuzz_module():
    """Drop-in modules on the GPU (forward -> elbo -> backward, the reference call pattern) vs autograd through the
    op-by-op CPU oracle in fp64, on random configurations: loss and every parameter gradient."""
    from vibo_amd.torch_core.models import VIBO_1PL, VIBO_2PL, VIBO_3PL
    t0, n, worst = time.time(), 0, 0.0
    while time.time() - t0 < a.seconds:
        irt = rng.choice([1, 2, 2, 3])
        A = rng.choice([1, 1, 2, 3, 4, 8])
        I = rng.choice([8, 95, 100, 256, 260, 1000, 1028, 2500])
        B = rng.choice([1, 8, 9, 17, 64])
        cond = rng.random() < 0.3
        n_flows = rng.choice([0, 0, 2, 4])
        drop = rng.random() < 0.3
        missing = rng.choice([0.0, 0.2])
        beta = rng.choice([1.0, 0.5])
        if irt == 3:          # 3PL: clamp-band chaos grows with the logit spread (see above): narrow abilities, no flows
            A, n_flows = min(A, 2), 0
        use_kl = n_flows == 0 and rng.random() < 0.7
        merge = 'mean' if (not cond and rng.random() < 0.3) else 'product'      # --ability-merge mean: unconditional only
        seed = rng.randrange(1 << 30)
        if a.replay:      # "irt A B I cond flows drop missing beta use_kl seed [merge]"
            f = a.replay.split()
            irt, A, B, I, n_flows, seed = int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[5]), int(f[10])
            cond, drop, use_kl, missing, beta = f[4] == 'True', f[6] == 'True', f[9] == 'True', float(f[7]), float(f[8])
            merge = f[11] if len(f) > 11 else 'product'
        torch.manual_seed(seed)
        cls = {1: VIBO_1PL, 2: VIBO_2PL, 3: VIBO_3PL}[irt]
        model = cls(A, I, hidden_dim=16, ability_merge=merge, conditional_posterior=cond,
                    replace_missing_with_prior=not drop, n_norm_flows=n_flows)
        with torch.no_grad():      # keep the logits out of the Bernoulli clamp band (fp32 decisions there are chaotic,
            # in the reference as well: the saturation golden pins the exact 1PL/2PL semantics separately)
            model.item_encoder.mu_lookup.weight.mul_(0.3)
            model.item_encoder.logvar_lookup.weight.mul_(0.3).sub_(2.0)
            for name, prm in model.named_parameters():       # small posterior means, well-conditioned flows: |logit| < ~8
                if name.endswith('mlp.4.weight') or name.endswith('mlp2.2.weight'):
                    prm.mul_(0.2)
                elif name.endswith('mlp2.2.bias'):           # mean merge: log-variances around -2 (the product of experts gets
                    prm[A:] = -2.0                           # its narrow posteriors from the number of experts)
                elif '_norm_flows' in name and name.endswith('.w'):      # uhat ~ w / |w|^2: keep |w| away from 0
                    prm.copy_(torch.sign(prm) * (0.5 + prm.abs()) / prm.numel() ** 0.5)
                elif '_norm_flows' in name and name.endswith('.u'):
                    prm.mul_(0.5)
        g = torch.Generator().manual_seed(seed + 1)
        resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=missing)
        if drop and missing > 0:
            mask[:, 0] = 1
            resp[:, 0] = resp[:, 0].clamp(min=0)
        eps_item = torch.randn(I, O.item_feat_dim(irt, A), generator=g)
        eps_ab = torch.randn(B, A, generator=g)
        cfg = dict(irt_model=irt, ability_dim=A, conditional_posterior=cond, replace_missing_with_prior=not drop,
                   n_norm_flows=n_flows, annealing_factor=beta, use_kl_divergence=use_kl)
        # fp32 like the reference: the Bernoulli probability clamp (eps of the dtype) is part of the semantics, an
        # fp64 oracle would clamp at |logit| = 36 instead of 15.94
        params = {k: v.detach().clone() for k, v in model.state_dict().items()}
        ref_out, ref_grads = O.elbo_loss_and_grads(params, resp, mask, eps_item, eps_ab, **cfg)
        ref_loss = ref_out['loss']
        model = model.to(d)
        outs = model(resp.to(d).unsqueeze(2), mask.to(d).bool().unsqueeze(2), eps_item=eps_item.to(d), eps_ability=eps_ab.to(d))
        if n_flows > 0:
            (r, k, rmu, ak, a0, amu, alv, aladj, ik, i0, imu, ilv, iladj) = outs
            loss = model.elbo(r, k, rmu, a0, amu, alv, i0, imu, ilv, annealing_factor=beta, use_kl_divergence=False,
                              ability_k=ak, item_feat_k=ik, ability_logabsdetjac=aladj, item_logabsdetjac=iladj)
        else:
            loss = model.elbo(*outs, annealing_factor=beta, use_kl_divergence=use_kl)
        loss.backward()
        # the fp32 oracle carries the reference's own noise (log(1-p) at |logit| ~ 10, clamp-band decisions), so this
        # sweep looks for gross host-logic errors: gradients are compared per parameter group, relative to the group's
        # largest entry
        errs = {'loss': rel(loss.detach().cpu(), ref_loss)}
        groups = {}
        for name, prm in model.named_parameters():
            grp = name.split('.')[0]
            e, m = groups.get(grp, (0.0, 0.0))
            groups[grp] = (max(e, float((prm.grad.cpu().double() - ref_grads[name].double()).abs().max())),
                           max(m, float(ref_grads[name].abs().max())))
        for grp, (e, m) in groups.items():
            if m > 1e-12:
                errs[grp] = e / m
        bad = {k: v for k, v in errs.items() if not (v < (5e-4 if k == 'loss' else 1e-2))}
        worst = max(worst, max(errs.values()))
        n += 1
        if a.replay:
            print('replayed module:', {k: float(f'{v:.3g}') for k, v in errs.items()})
            sys.exit(0)
        if bad:
            print(f'FAIL module irt={irt} A={A} B={B} I={I} cond={cond} flows={n_flows} drop={drop} missing={missing} beta={beta} '
                  f'use_kl={use_kl} seed={seed} merge={merge}: {bad}')
            sys.exit(1)
    print(f'fuzz module ok: {n} random configurations, worst relative error {worst:.2e}')


def fuzz_trainer():
    """FusedTrainer (prologue / fused ELBO / epilogue+Adam kernels) vs module + autograd + torch.optim.Adam under the same
    noise, 3 steps: losses and every parameter after the steps."""
    import copy
    from vibo_amd.torch_core.models import VIBO_1PL, VIBO_2PL, VIBO_3PL
    from vibo_amd.trainer import FusedTrainer
    t0, n, worst, n_band, n_arb = time.time(), 0, 0.0, 0, 0
    while time.time() - t0 < a.seconds:
        irt = rng.choice([1, 2, 2, 3])
        A = rng.choice([1, 2, 3, 5, 8])
        I = rng.choice([4, 37, 95, 100, 255, 256, 257, 1000, 1028])
        B = rng.choice([1, 9, 64, 130])
        hidden = rng.choice([4, 16, 33, 64, 100, 256])
        beta = rng.choice([1.0, 0.5, 0.0])
        lr = rng.choice([5e-3, 1e-2])
        gather = rng.random() < 0.3
        cls = {1: VIBO_1PL, 2: VIBO_2PL, 3: VIBO_3PL}[irt]
        seed = rng.randrange(1 << 30)
        g = torch.Generator().manual_seed(seed)
        P = B + 40 if gather else B
        resp, mask = O.simulate_responses(irt, P, I, A, generator=g, missing_frac=rng.choice([0.0, 0.15]))
        rows = torch.randint(0, P, (B,), generator=g).to(d) if gather else None
        resp, mask = ops.pad_rows(resp.to(d), mask.bool().to(d))
        torch.manual_seed(seed)
        ref = cls(A, I, hidden_dim=hidden, ability_merge='product').to(d)
        if irt == 3:      # keep 3PL logits out of the probability-clamp band (|logit| > ~12), where 1-ulp differences of the
            with torch.no_grad():      # item sample (torch ops vs the prologue kernel) flip single cells with O(1) gradients
                ref.item_encoder.mu_lookup.weight.mul_(0.5 if A <= 3 else 0.25)      # (the logit's spread grows with sqrt(A): seed 777 found
                #  3PL, 8 dims, 256 items at 1.2 % of the encoder's first moment with 0.5 -- tests/test_gpu_trainer.py notes the same band)
        fus = copy.deepcopy(ref)
        # 3PL: cells of the first step whose probability (float64, from the module's own samples) sits within a few fp32 ulps of
        # the clamp bound 1 - eps32 -- where a 1-ulp difference of the item sample decides whether the cell's O(1) gradient
        # counts.  A report with band == 0 is a real discrepancy; with band > 0 it is that class.
        band = 0
        if irt == 3:
            torch.manual_seed(100)
            with torch.no_grad():
                out0 = ref.forward(resp, mask, row_index=rows)
            pr = O.irt_link(irt, out0[3].double().cpu(), out0[6].double().cpu())
            mk = ((mask[rows] if rows is not None else mask)[:, :pr.shape[1]] != 0).cpu()
            e32 = 1.1920929e-07
            band = int(((pr > 1.0 - 5 * e32) & mk).sum()) + int(((pr < 5 * e32) & mk).sum())
            n_band += band > 0
        opt = torch.optim.Adam(ref.parameters(), lr=lr)
        trainer = FusedTrainer(fus, lr=lr)
        for step in range(3):
            torch.manual_seed(100 + step)
            opt.zero_grad()
            loss_ref = ref.elbo_step(resp, mask, annealing_factor=beta, row_index=rows)
            loss_ref.backward()
            opt.step()
            torch.manual_seed(100 + step)
            loss_fus = trainer.step(resp, mask, beta=beta, row_index=rows)
            e = abs(float(loss_fus) - float(loss_ref.detach())) / max(1.0, abs(float(loss_ref.detach())))
            worst = max(worst, e) if step == 0 else worst
            if step == 0:
                # after ONE step the first moments are 0.1 x gradient: the sharp check of prologue / kernel / epilogue
                # (later steps see Adam's m / sqrt(v) amplify 1e-7 parameter differences chaotically)
                mlp = ref.ability_encoder.mlp
                ref_m = torch.cat([opt.state[p_]['exp_avg'].reshape(-1) for p_ in (mlp[0].weight, mlp[0].bias, mlp[2].weight,
                                                                                 mlp[2].bias, mlp[4].weight, mlp[4].bias)])
                ref_im = torch.cat([opt.state[ref.item_encoder.mu_lookup.weight]['exp_avg'].reshape(-1),
                                    opt.state[ref.item_encoder.logvar_lookup.weight]['exp_avg'].reshape(-1)])
                for name, x, y in (('mlp exp_avg', trainer.mlp_m, ref_m), ('item exp_avg', trainer.item_m, ref_im)):
                    em = float((x - y).abs().max()) / max(1e-1, float(y.abs().max()))     # (0.1 g; vanishing gradients: absolute floor)
                    worst = max(worst, em)
                    if not em < 1e-4 and name == 'mlp exp_avg' and irt != 3 and em < 2e-3:
                        # two fp32 paths disagree on the encoder's first moment: the fp64 oracle under the SAME noise decides (the noise the
                        # module drew, recovered from a forward of a freshly constructed -- seeded -- module).  Round 6, seed 645: hidden
                        # 256 at beta 0 (nothing but the log-likelihood feeds the encoder): torch's fp32 MLP backward 1.1e-4 off the fp64
                        # gradient, the native epilogue 4.7e-5, 1.6e-4 between them.
                        torch.manual_seed(seed)
                        fresh = cls(A, I, hidden_dim=hidden, ability_merge='product').to(d)
                        torch.manual_seed(100)
                        with torch.no_grad():
                            o = fresh.forward(resp, mask, row_index=rows)
                        e_ab = ((o[3] - o[4]) / torch.exp(0.5 * o[5])).cpu().double()
                        e_it = ((o[6] - o[7]) / torch.exp(0.5 * o[8])).cpu().double()
                        rr, mm = (resp[rows], mask[rows]) if rows is not None else (resp, mask)
                        names = [f'ability_encoder.mlp.{k_}.{w_}' for k_ in (0, 2, 4) for w_ in ('weight', 'bias')]
                        _, g64 = O.elbo_loss_and_grads({k_: v_.detach().cpu().double() for k_, v_ in fresh.state_dict().items()}, rr[:, :I].cpu().double(),
                                                       mm[:, :I].cpu(), e_it, e_ab, irt_model=irt, ability_dim=A, annealing_factor=beta)
                        t64 = torch.cat([g64[n_].reshape(-1) for n_ in names]) * 0.1
                        e64 = float((x.cpu().double() - t64).abs().max()) / max(1e-1, float(t64.abs().max()))
                        e64m = float((y.cpu().double() - t64).abs().max()) / max(1e-1, float(t64.abs().max()))
                        print(f'  (mlp exp_avg, module vs native step {em:.2e} at irt={irt} A={A} B={B} I={I} hidden={hidden} beta={beta} seed={seed}: against the fp64 '
                              f'oracle the native step is off by {e64:.2e}, the module path by {e64m:.2e})')
                        n_arb += 1
                        em = e64
                    if not em < 1e-4:
                        print(f'FAIL trainer {name} irt={irt} A={A} B={B} I={I} hidden={hidden} beta={beta} lr={lr} gather={gather} seed={seed}: {em}  (cells in the clamp band: {band})')
                        sys.exit(1)
            if not e < (5e-5 if step == 0 else 3e-2):
                print(f'FAIL trainer loss irt={irt} A={A} B={B} I={I} hidden={hidden} beta={beta} lr={lr} gather={gather} seed={seed} step={step}: {e}')
                sys.exit(1)
        for (k, x), (_, y) in zip(ref.state_dict().items(), fus.state_dict().items()):
            e = float((x - y).abs().max())
            if not e < 6.0 * lr:      # boundedness only (3 steps x lr each way): later steps are chaotic, see above
                print(f'FAIL trainer param {k} irt={irt} A={A} B={B} I={I} hidden={hidden} beta={beta} lr={lr} gather={gather} seed={seed}: {e}')
                sys.exit(1)
        n += 1
    print(f'fuzz trainer ok: {n} random configurations, worst error {worst:.2e} ({n_band} of them 3PL with cells in the clamp band; {n_arb} first-moment disagreements of the two fp32 paths settled by the fp64 oracle)')


if a.target == 'multi':
    fuzz_multi()
    sys.exit(0)
if a.target == 'trainer':
    fuzz_trainer()
    sys.exit(0)
if a.target == 'module':
    fuzz_module()
    sys.exit(0)

t0, n, worst = time.time(), 0, 0.0
n_band3 = 0
while time.time() - t0 < a.seconds:
    irt = rng.choice([1, 2, 2, 3])
    A = rng.choice([1, 1, 2, 3, 4, 5, 8, 8, 9, 12, 16])      # (9..16: the wave-per-person kernel's wide instantiation)
    I = rng.choice([4, 8, 64, 96, 100, 255, 256, 257, 260, 511, 512, 516, 768, 1000, 1023, 1024, 1028, 2048, 2500, 3000])
    B = rng.choice([1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 130, 257])
    cond = rng.random() < 0.25
    n_flows = rng.choice([0, 0, 0, 1, 4, 8])
    drop = rng.random() < 0.3
    missing = rng.choice([0.0, 0.1, 0.5])
    pad = rng.random() < 0.7
    # large item parameters drive logits into the Bernoulli clamp: exact for 1PL/2PL (tested separately against the
    # saturation golden); for 3PL the clamp acts on p itself and the fp32 / fp64 decision differs in a narrow band
    # where single cells carry O(1) gradients, so keep 3PL and wide abilities away from it
    scale = rng.choice([0.5, 1.0, 3.0]) if (irt != 3 and A <= 2) else rng.choice([0.3, 0.6])
    dataseed = rng.randrange(1 << 30)
    gather = rng.random() < 0.3          # minibatch as a row-index vector over a larger resident matrix
    no_mask = missing == 0.0 and rng.random() < 0.3
    fwd_only = rng.random() < 0.2
    codes = rng.random() < 0.35 and not no_mask and not (cond and A > 4) and A <= 8      # rows as 1-byte cell codes (Format P)
    given = (not cond) and A <= 8 and rng.random() < 0.25      # caller-supplied posterior (VIBO_POSTERIOR_GIVEN, --ability-merge mean)
    # which row-split kernel: the planner's choice (the narrow-row kernel for <= 128 items at ability_dim <= 4, else by size), or one
    # pinned through vibo_desc.flags -- at these minibatch sizes the planner alone never picks the matrix kernel (round 5)
    kflag = rng.choice([0, 0, _lib.FLAG_KERNEL_MATRIX | _lib.FLAG_COND_MATRIX, _lib.FLAG_KERNEL_VALU | _lib.FLAG_COND_VALU])
    if a.focus == 'cond1':
        A, cond, n_flows, codes, given = 1, True, 0, False, False
        B = rng.choice([1, 9, 31, 32, 33, 64, 65, 257, 1000, 4099])
        I = rng.choice([4, 8, 96, 100, 127, 128, 129, 255, 257, 512, 516, 768, 897, 1000, 1023, 1024])
        kflag = _lib.FLAG_KERNEL_MATRIX | _lib.FLAG_COND_MATRIX
    if a.replay:
        f = a.replay.split()
        irt, A, B, I, n_flows, dataseed = int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[5]), int(f[10])
        cond, drop, pad, missing, scale = f[4] == 'True', f[6] == 'True', f[8] == 'True', float(f[7]), float(f[9])
        gather, no_mask, fwd_only = f[11] == 'True', f[12] == 'True', f[13] == 'True'
        codes = len(f) > 14 and f[14] == 'True'
        given = len(f) > 15 and f[15] == 'True'
        kflag = int(f[16]) if len(f) > 16 else 0
    if given:
        pad = True                 # this mode exists on the row-split path only: rows padded like the resident data path does
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=cond, drop_missing=drop, n_flows=n_flows, given=given)
    g = torch.Generator().manual_seed(dataseed)
    P = B + rng.choice([3, 50]) if gather else B
    resp_all, mask_all = O.simulate_responses(irt, P, I, A, generator=g, missing_frac=missing)
    rows = torch.randint(0, P, (B,), generator=g) if gather else None
    resp, mask = (resp_all[rows], mask_all[rows]) if gather else (resp_all, mask_all)
    if drop and missing > 0:
        mask_all[:, 0] = 1
        resp_all[:, 0] = resp_all[:, 0].clamp(min=0)
        resp, mask = (resp_all[rows], mask_all[rows]) if gather else (resp_all, mask_all)
    D = O.item_feat_dim(irt, A)
    table = torch.randn((2, I, 2 * A) if cond else (2, 2 * A), generator=g) * 0.7
    if given:                      # (mu | logvar) per person
        # (3PL: narrower, so that flows on wide samples do not push logits into the probability-clamp band, see above)
        table = torch.cat([torch.randn(B, A, generator=g) * (0.3 if irt == 3 else 0.8),
                           torch.randn(B, A, generator=g) * 0.6 - (3.0 if irt == 3 else 1.5)], dim=1)
    item = torch.randn(I, D, generator=g) * scale
    eps = torch.randn(B, A, generator=g)
    flow = None
    if n_flows:      # (u, w, b) as the model holds them; the kernel receives uhat (flows.py:23-25), which keeps w.uhat > -1
        raw_f = torch.randn(n_flows, 2 * A + 1, generator=g) * 0.7
        # |w| bounded away from 0: uhat ~ w / |w|^2 explodes otherwise, theta_K reaches the hundreds and every logit sits
        # in the Bernoulli clamp (3PL: fp32-vs-fp64 clamp decisions, see the item-scale note above)
        wv = raw_f[:, A:2 * A]
        raw_f[:, A:2 * A] = torch.sign(wv) * (0.4 + wv.abs()) / (A ** 0.5)
        flow = torch.stack([torch.cat([T.flow_uhat(f[:A], f[A:2 * A]), f[A:]]) for f in raw_f])
    flows = [(f[:A].double(), f[A:2 * A].double(), f[2 * A:].double()) for f in flow] if n_flows else None
    mode = 'sampled' if n_flows else 'kl'
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(), irt_model=irt, ability_dim=A,
                           conditional_posterior=cond, replace_missing_with_prior=not drop, mode=mode, flow_uhat_w_b=flows,
                           given_posterior=given)
    r_, m_ = (ops.pad_rows(resp_all.to(d), mask_all.bool().to(d)) if pad else (resp_all.to(d), mask_all.bool().to(d)))
    if codes:
        r_, m_ = ops.pack_cell_codes(resp_all.to(d), mask_all.bool().to(d)), None
    r, m, code = ops.prepare_rows(r_, None if no_mask else m_)
    ops.DESC_FLAGS = kflag
    raw = ops._hip_launch_elbo(spec, r, m, code, rows.to(d) if gather else None, table.to(d).contiguous(), item.to(d).contiguous(),
                               eps.to(d).contiguous(), flow.to(d).contiguous() if flow is not None else None,
                               _lib.REG_SAMPLED if n_flows else _lib.REG_KL, not fwd_only, B)
    torch.cuda.synchronize()
    sc = raw.scalars.cpu()
    errs = {
        # (relative to |ll| with a floor of 0.02 per cell of the call's 32-row batches: a one-person call whose few cells are
        #  all well predicted has |ll| ~ 1 while its fp32 sum runs over thousands of padding cells)
        'll': abs(float(sc[_lib.S_LL]) - float(ref['ll'])) / max(abs(float(ref['ll'])), 0.02 * ((B + 31) // 32 * 32) * I),
        'reg': abs(float(sc[_lib.S_REG]) - float(ref['reg'])) / max(1.0, abs(float(ref['reg']))),
        'mu': float((raw.ability_mu.cpu() - ref['ability_mu'].float()).abs().max()) / max(1.0, float(ref['ability_mu'].abs().max())),
        'theta': float((raw.ability.cpu() - ref['ability'].float()).abs().max()) / max(1.0, float(ref['ability'].abs().max())),
    }
    if not fwd_only:
        errs['g_item'] = rel(raw.grad_item((I, D)).cpu(), ref['g_item'])
    for s_ in range(0 if fwd_only else 2):
        if float(ref['g_table'][s_].abs().max()) > 0:
            # (relative to the largest entry, with an absolute floor: a single person's d LL / d table can cancel to ~1e-5,
            # where fp32 noise of 1e-7 is not a defect)
            gt_ref = ref['g_table'][s_]
            errs[f'g_table{s_}'] = float((raw.grad_table(s_).cpu().double() - gt_ref.double()).abs().max()) / max(1e-2, float(gt_ref.abs().max()))
    if n_flows and not fwd_only:
        for s_ in range(2):
            gref = torch.cat([torch.cat(gf) for gf in ref['g_flow'][s_]]).float()
            if float(gref.abs().max()) > 0:
                errs[f'g_flow{s_}'] = rel(raw.grad_flow(s_).cpu(), gref)
    # model.encode's kernel on the same rows (row-statistics fast path or wave-per-person fallback)
    emu, elv = (raw.ability_mu, raw.ability_logvar) if given else ops.encode_posterior(spec, table.to(d), r_, None if no_mask else m_, row_index=rows.to(d) if gather else None)
    errs['enc_mu'] = float((emu.cpu() - ref['ability_mu'].float()).abs().max()) / max(1.0, float(ref['ability_mu'].abs().max()))
    errs['enc_lv'] = float((elv.cpu() - ref['ability_logvar'].float()).abs().max()) / max(1.0, float(ref['ability_logvar'].abs().max()))
    lim = {'ll': 3e-5, 'reg': 3e-5, 'mu': 3e-5, 'theta': 6e-5, 'enc_mu': 3e-5, 'enc_lv': 3e-5}
    bad = {k: v for k, v in errs.items() if not (v < lim.get(k, 6e-4))}
    if bad and irt != 3 and not fwd_only and float(ref['logit'].abs().max()) > 15.9 and set(bad) <= {'g_item', 'g_table0', 'g_flow0'}:
        # Logits inside the Bernoulli clamp band: the fp64 reference zeroes the gradient of a cell for |l| > 15.94 on both sides,
        # the reference's own fp32 arithmetic (which the kernel follows) rounds sigmoid(l) to 1 - 2^-24 k and clamps the upper
        # side later; a badly predicted cell there carries an O(1) gradient.  Decide such a case against the SAME op sequence
        # in fp32.
        ref32 = T.fused_elbo_ref(table.float(), item.float(), resp.float(), mask, eps.float(), irt_model=irt, ability_dim=A,
                                 conditional_posterior=cond, replace_missing_with_prior=not drop, mode=mode,
                                 flow_uhat_w_b=[tuple(t.float() for t in f3) for f3 in flows] if flows else None,
                                 given_posterior=given)
        e32 = {'g_item': rel(raw.grad_item((I, D)).cpu(), ref32['g_item'].float()),
               'g_table0': float((raw.grad_table(0).cpu().double() - ref32['g_table'][0].double()).abs().max())
               / max(1e-2, float(ref32['g_table'][0].abs().max()))}
        if n_flows:
            e32['g_flow0'] = rel(raw.grad_flow(0).cpu(), torch.cat([torch.cat(gf) for gf in ref32['g_flow'][0]]).float())
        bad = {k: v for k, v in e32.items() if not (v < 6e-4)}
        if a.replay:
            print('against the fp32 op sequence (clamp band):', e32)
    band3 = False
    if bad and irt == 3 and not fwd_only and not a.replay and float(ref['logit'].abs().max()) > 15.9 and set(bad) <= {'g_item', 'g_table0', 'g_flow0'}:
        # 3PL with cells inside the clamp band (flows that push the sample out: |logit| up to 18): the clamp acts on
        # p = guess + (1 - guess) sigmoid(l) and flips with the last bit of that sum -- in the reference's own fp32 arithmetic as much
        # as here (known limits); neither fp64 nor an fp32 restatement arbitrates a single cell's O(1) gradient.  Counted
        # and reported, not a failure: round 6's two reports of this class reproduce digit for digit on round 5's library.
        n_band3 += 1
        band3 = True
        print(f'3PL clamp-band case (counted): A={A} B={B} I={I} flows={n_flows} given={given} kflag={kflag} max|logit|={float(ref["logit"].abs().max()):.2f} {bad}')
        print(f'  replay: --replay "{irt} {A} {B} {I} {cond} {n_flows} {drop} {missing} {pad} {scale} {dataseed} {gather} {no_mask} {fwd_only} {codes} {given} {kflag}"')
        bad = {}
    if not band3:
        worst = max(worst, max(errs.values()))
    n += 1
    if a.replay:
        print('ll', float(sc[_lib.S_LL]), 'ref', float(ref['ll']))
        print('replayed:', errs, '| max |logit| of the case:', float(ref['logit'].abs().max()),
              '| max |g_table| per set:', [float(t.abs().max()) for t in ref.get('g_table', [])],
              '| d LL / d theta range:', (float(ref['g_item'].abs().max()) if 'g_item' in ref else None))
        sys.exit(1 if bad else 0)
    if bad:
        print(f'FAIL irt={irt} A={A} B={B} I={I} cond={cond} flows={n_flows} drop={drop} missing={missing} pad={pad} gather={gather} no_mask={no_mask} fwd_only={fwd_only} codes={codes} given={given} kflag={kflag}: {bad}')
        print(f'replay: --replay "{irt} {A} {B} {I} {cond} {n_flows} {drop} {missing} {pad} {scale} {dataseed} {gather} {no_mask} {fwd_only} {codes} {given} {kflag}"')
        sys.exit(1)
print(f'fuzz ok: {n} random configurations, worst relative error {worst:.2e} ({n_band3} 3PL clamp-band cases counted apart)')
