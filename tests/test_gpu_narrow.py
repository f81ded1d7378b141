"""GPU parity of the narrow-row kernel (csrc/vibo_narrow.hip: 4..128 items, ability_dim <= 4, plain model -- BASELINE
configs[0]'s 100 items and configs[3]'s CritLangAcq rows of 95; the planner's choice whenever no flag pins a row-split kernel):
 (1) the reference's goldens of those shapes through the module and through the native train step,
 (2) the CPU oracle on seeded random problems: every link, ragged item counts with padded strides, row_index gather, cell codes,
     no mask, --drop-missing, rows past the last whole unit, single items / persons,
 (3) the reference's saturation probe (gradient exactly zero where the reference's is),
 (4) size-independent properties at config 3's full size (shard additivity, permutation invariance, bitwise determinism).

The other GPU modules pin one of the row-split kernels per fixture run (vibo_desc.flags), so their shapes never reach this
kernel; here ops.DESC_FLAGS stays 0 and every case asserts that the planner picked it.
Tolerances: ELBO <= 1e-4 relative, posterior <= 2e-5, gradients <= 1e-4 of the tensor's max-abs.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, rel_err
from oracle import vibo_table_ref as T
from test_gpu_parity import (compare_raw, dev, random_problem, TOL_ELBO, _device_problem)
from test_host_logic import build_model, check_against_golden, run_reference_pattern
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

pytestmark = pytest.mark.gpu

NARROW = 'narrow rows (narrow_kernel)'


def is_narrow(spec, B, I, mcode=_lib.MASK_U8, grad=True):
    return ops.plan_kernel(spec, B, I, mcode, grad) == NARROW


def launch(spec, resp, mask, table, item, eps, *, row_index=None, codes=False, want_grad=True):
    """resp / mask on the host; rows padded to 16-byte strides (ops.pad_rows) as the CLI's resident splits are."""
    d = dev()
    assert ops.DESC_FLAGS == 0
    if codes:
        cells = ops.pack_cell_codes(resp.to(d), mask.to(d).bool() if mask is not None else None).codes
        r, m, code = cells, cells, _lib.MASK_CODES
    else:
        rp, mp = ops.pad_rows(resp.to(d), mask.to(d).bool() if mask is not None else None)
        r, m, code = ops.prepare_rows(rp, mp)
    ri = row_index.to(d) if row_index is not None else None
    B = int(ri.numel()) if ri is not None else resp.shape[0]
    assert is_narrow(spec, B, resp.shape[1], code, want_grad), 'planner did not pick the narrow-row kernel'
    raw = ops._hip_launch_elbo(spec, r, m, code, ri, table.to(d).contiguous(), item.to(d).contiguous(), eps.to(d).contiguous(),
                               None, _lib.REG_KL, want_grad, B)
    torch.cuda.synchronize()
    return raw


SHAPES = [
    # irt, A, B, I, missing
    (2, 1, 31, 95, 0.2),          # CritLangAcq's width: padded row strides, 12 of 16 lanes
    (2, 1, 8000, 100, 0.0),       # BASELINE configs[0]'s train split
    (2, 1, 1000, 100, 0.3),
    (2, 2, 300, 96, 0.1),
    (2, 4, 257, 128, 0.2),        # widest row, widest posterior
    (2, 3, 70, 100, 0.3),         # ability_dim padded 3 -> 4
    (2, 1, 129, 64, 0.1),         # 4 items per lane, every lane busy
    (2, 2, 33, 36, 0.2),
    (2, 4, 9, 8, 0.0),
    (2, 1, 5, 4, 0.0),            # one chunk per row
    (2, 2, 1, 7, 0.0),            # single person, ragged
    (2, 1, 7, 65, 0.5),           # first width of the 8-items-per-lane instantiation
    (1, 1, 100, 64, 0.1),
    (1, 4, 77, 36, 0.2),
    (1, 2, 400, 100, 0.0),
    (3, 1, 100, 100, 0.1),
    (3, 2, 50, 95, 0.3),
    (3, 4, 64, 128, 0.0),
    (3, 1, 333, 40, 0.2),
]


@pytest.mark.parametrize('irt,A,B,I,missing', SHAPES)
@pytest.mark.parametrize('drop', [False, True])
def test_narrow_kernel_vs_oracle(irt, A, B, I, missing, drop):
    spec = ElboSpec(irt_model=irt, ability_dim=A, drop_missing=drop)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, missing, seed=B * 7 + I + A)
    if drop and missing > 0:
        mask[:, 0] = 1
        resp[:, 0] = resp[:, 0].clamp(min=0)
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(), irt_model=irt, ability_dim=A,
                           replace_missing_with_prior=not drop, mode='kl')
    a = launch(spec, resp, mask, table, item, eps)
    compare_raw(a, ref, (I, spec.item_dim))
    b = launch(spec, resp, mask, table, item, eps)
    assert torch.equal(a.flat, b.flat) and torch.equal(a.ability_mu, b.ability_mu), 'must be bitwise deterministic'
    # forward only: the same posterior and heads
    f = launch(spec, resp, mask, table, item, eps, want_grad=False)
    assert torch.equal(f.ability_mu, a.ability_mu) and torch.equal(f.ability, a.ability)
    assert rel_err(f.scalars[_lib.S_LL], a.scalars[_lib.S_LL]) < 1e-6


@pytest.mark.parametrize('irt,A,B,I', [(2, 1, 200, 95), (3, 2, 77, 100), (1, 4, 130, 64), (2, 4, 41, 128)])
def test_narrow_kernel_row_modes(irt, A, B, I):
    """The three row modes of the kernel read the same cells: fp32 rows through a row_index (shuffled minibatch), 1-byte cell
    codes (Format P) in order and gathered -- bit for bit the fp32 in-order call on the same rows; and a matrix without a mask."""
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    resp, mask, table, item, eps = random_problem(irt, A, B + 19, I, 0.2, seed=I * A + B)
    g = torch.Generator().manual_seed(B)
    rows = torch.randperm(B + 19, generator=g)[:B]
    base = launch(spec, resp[rows], mask[rows], table, item, eps[:B])
    ref = T.fused_elbo_ref(table.double(), item.double(), resp[rows].double(), mask[rows], eps[:B].double(), irt_model=irt,
                           ability_dim=A, mode='kl')
    compare_raw(base, ref, (I, spec.item_dim))
    for kw in (dict(row_index=rows), dict(codes=True, row_index=rows)):
        out = launch(spec, resp, mask, table, item, eps[:B], **kw)
        assert torch.equal(out.flat, base.flat) and torch.equal(out.ability_mu, base.ability_mu), kw
    out = launch(spec, resp[rows], mask[rows], table, item, eps[:B], codes=True)
    assert torch.equal(out.flat, base.flat)
    # no mask at all (VIBO_MASK_NONE): every cell observed
    full = torch.ones_like(mask)
    a = launch(spec, resp[rows].clamp(min=0), None, table, item, eps[:B])
    b = launch(spec, resp[rows].clamp(min=0), full[rows], table, item, eps[:B])
    assert torch.equal(a.flat, b.flat)


def test_narrow_kernel_saturation_golden():
    """The reference's saturation probe (tests/golden/saturation.npz: one person at theta = 0, difficulties on a grid through
    the Bernoulli clamp, utils.py:46-49) in rows of 100 items: the gradient is exactly zero where the reference's is."""
    z = np.load(os.path.join(GOLDEN_DIR, 'saturation.npz'))
    logit_all = torch.from_numpy(z['logit'])
    spec = ElboSpec(irt_model=2, ability_dim=1)
    table = torch.zeros(2, 2)
    for s0 in range(0, 2000, 500):
        sel = slice(s0, s0 + 100)
        logit = logit_all[sel]
        I = logit.numel()
        item = torch.stack([torch.ones(I), logit], dim=1)      # a_i = 1, b_i = logit; table = 0, eps = 0 => theta = 0 exactly
        for x in (0, 1):
            resp = torch.full((1, I), float(x))
            raw = launch(spec, resp, torch.ones(1, I, dtype=torch.bool), table, item, torch.zeros(1, 1))
            g_b = raw.grad_item((I, 2))[:, 1].cpu()
            ref_g = torch.from_numpy(z[f'dll_dlogit_x{x}'])[sel]
            assert torch.equal(g_b == 0, ref_g == 0)
            assert (g_b - ref_g).abs().max() < 2e-6
            lc = logit.double().clamp(-T.LOGIT_LO, T.LOGIT_LO)
            ll_exact = float((x * lc - lc.clamp(min=0) - torch.log1p(torch.exp(-lc.abs()))).sum())
            assert abs(float(raw.scalars[_lib.S_LL]) - ll_exact) < 1e-5 * max(1.0, abs(ll_exact))


def _narrow_golden(golden):
    m = golden.meta
    I = golden.response.shape[1]
    return (4 <= I <= 128 and m['ability_dim'] <= 4 and m['n_norm_flows'] == 0 and not m['conditional_posterior'] and
            m.get('ability_merge', 'product') == 'product' and m.get('generative_model', 'irt') == 'irt')


def test_narrow_goldens_through_module(golden):
    """Every reference golden of a narrow plain model (tools/gen_golden.py; incl. BASELINE configs[0]'s 16 x 100 minibatch)
    through the drop-in module with the planner's own choice: loss, posterior, every parameter gradient."""
    if not _narrow_golden(golden):
        pytest.skip('not a narrow plain-model case')
    d = dev()
    model = build_model(golden).to(d)
    golden.response, golden.mask = golden.response.to(d), golden.mask.to(d)
    golden.eps_item, golden.eps_ability = golden.eps_item.to(d), golden.eps_ability.to(d)
    outs, loss = run_reference_pattern(model, golden)
    check_against_golden(model, golden, outs, loss, tol_loss=TOL_ELBO, tol_grad=1e-4, strict=True)


def test_narrow_goldens_adam_trajectory_through_the_fused_trainer(golden):
    """... and the reference's recorded 1- and 3-step Adam parameters through the folded native train step on this kernel."""
    from vibo_amd.trainer import FusedTrainer, fused_trainer_covers
    if not _narrow_golden(golden):
        pytest.skip('not a narrow plain-model case')
    m = golden.meta
    model = build_model(golden)
    if not fused_trainer_covers(model) or not m['use_kl_divergence']:
        pytest.skip('configuration trains through the module path')
    d = dev()
    model = model.to(d)
    tr = FusedTrainer(model, lr=5e-3)
    resp, mask = ops.pad_rows(golden.response.to(d), golden.mask.to(d).bool())
    assert is_narrow(model.spec, resp.shape[0], resp.shape[1])
    eps_i, eps_a = golden.eps_item.to(d), golden.eps_ability.to(d)
    for step in range(3):
        loss = tr.step(resp, mask, beta=m['annealing_factor'], eps_item=eps_i, eps_ability=eps_a)
        if step == 0:
            assert rel_err(loss, golden.out['loss']) < TOL_ELBO
            for k, v in golden.adam1.items():
                assert (model.state_dict()[k].cpu() - v).abs().max() < 2e-4, (k, 'after one step')
    for k, v in golden.adam3.items():
        assert (model.state_dict()[k].cpu() - v).abs().max() < 5e-4, k


def test_narrow_kernel_config4_shape_at_full_size():
    """BASELINE configs[3]'s matrix shape at full size (535 598 x 95, 2PL, ability_dim 1, 20 % missing; datasets.py:283-440,
    models.py:596-629) on the narrow-row kernel: bitwise reproducible, additive over two person shards, invariant under a person
    permutation through the in-kernel gather, equal to the CPU oracle on a slice, and within fp32 noise of the VALU row-split
    kernel on the whole matrix."""
    irt, A, P, I = 2, 1, 535_598, 95
    d = dev()
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    assert is_narrow(spec, P, I)
    resp, mask, table, item, eps = _device_problem(irt, A, P, I, 0.2, seed=44, cond=False)
    rp, mp = ops.pad_rows(resp, mask)

    def run(rows=None, row_index=None):
        r = rp if rows is None else rp[rows]
        m = mp if rows is None else mp[rows]
        e = eps if rows is None else eps[rows].contiguous()
        if row_index is not None:
            e = eps[row_index].contiguous()
        r2, m2, code = ops.prepare_rows(r, m)
        B = int(row_index.numel()) if row_index is not None else r.shape[0]
        out = ops._hip_launch_elbo(spec, r2, m2, code, row_index, table, item, e, None, _lib.REG_KL, True, B)
        torch.cuda.synchronize()
        return out

    full, again = run(), run()
    assert torch.equal(full.flat, again.flat) and torch.equal(full.ability_mu, again.ability_mu)
    h = P // 2 + 37
    a, b = run(slice(0, h)), run(slice(h, P))
    summed = a.flat + b.flat
    assert rel_err(summed[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(summed[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    assert torch.equal(torch.cat([a.ability_mu, b.ability_mu]), full.ability_mu)
    perm = torch.randperm(P, device=d, generator=torch.Generator(device=d).manual_seed(9))
    pg = run(row_index=perm)
    assert rel_err(pg.flat[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(pg.flat[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    assert torch.equal(pg.ability_mu, full.ability_mu[perm])
    n = 512
    ref = T.fused_elbo_ref(table.cpu().double(), item.cpu().double(), resp[:n].cpu().double(), mask[:n].cpu(),
                           eps[:n].cpu().double(), irt_model=irt, ability_dim=A, mode='kl')
    compare_raw(run(slice(0, n)), ref, (I, A + 1))
    with ops.desc_flags(_lib.FLAG_KERNEL_VALU):
        v = run()
    assert rel_err(v.flat[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(v.flat[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    assert (v.ability_mu - full.ability_mu).abs().max() < 1e-6
