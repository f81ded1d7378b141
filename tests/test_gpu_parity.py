"""GPU parity: the fused HIP kernel (called through the C ABI) against
 (1) the committed golden vectors generated from the real reference,
 (2) the CPU oracle on seeded random problems, incl. ragged / odd shapes, every
     mask dtype, in-kernel row gather, all-missing rows, saturated logits,
 (3) size-independent properties at BASELINE.json sizes (shard additivity,
     person-permutation invariance, bitwise determinism).

Tolerances (fp32): ELBO <= 1e-4 relative (north_star), posterior
mean / log-variance <= 2e-5, gradients <= 1e-4 of the tensor's max-abs against the fp64 analytic oracle (TOL_GRAD; measured
maximum over the suite 6.1e-6) -- goldens: see check_against_golden.
"""
import json
import os

import pytest
import torch

from conftest import rel_err
from oracle import vibo_oracle as O
from oracle import vibo_table_ref as T
from test_host_logic import build_model, check_against_golden, run_reference_pattern
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[_lib.FLAG_KERNEL_MATRIX | _lib.FLAG_COND_MATRIX, _lib.FLAG_KERNEL_VALU | _lib.FLAG_COND_VALU,
                        _lib.FLAG_KERNEL_MATRIX | _lib.FLAG_NO_EMIT_CODES | _lib.FLAG_COND_VALU, _lib.FLAG_KERNEL_VALU,
                        _lib.FLAG_KERNEL_MATRIX | _lib.FLAG_COND_MATRIX | _lib.FLAG_COND_THREE_PASS],
                ids=['matrix-kernels', 'valu-kernels', 'matrix-kernel-fp32-passes', 'valu-kernel-planned-posterior', 'matrix-kernels-three-pass'])
def row_split_kernel_choice(request, monkeypatch):
    """Every test here runs on both row-split kernels: the library's planner picks the matrix kernel (vibo_msplit_kernel.hpp)
    above 2 048 persons per call and the VALU kernel (vibo_split_kernel.hpp) below; vibo_desc.flags pins one for the whole test
    (ops.DESC_FLAGS: the library reads no environment variable).  Third run: the multi-pass paths (conditional posterior, more
    than 1024 items) re-read the fp32 rows in every pass instead of the 1-byte cell codes their first pass leaves behind
    (VIBO_FLAG_NO_EMIT_CODES).  The conditional posterior's two passes have a matrix-pipe form (vibo_cmean.hip, the default
    from a call size that depends on ability_dim when the rows are cell codes, VIBO_FLAG_COND_MATRIX pins it) and a VALU form
    (vibo_cond.hip, VIBO_FLAG_COND_VALU): the first run pins the matrix-pipe form, the second and third the VALU form, the fourth
    runs the VALU row-split kernel around whatever the planner picks.  Fifth run (tests of the conditional posterior only): at
    ability_dim 1 on fp32 rows the first run's matrix kernel gathers the experts itself (its XM == 3); VIBO_FLAG_COND_THREE_PASS keeps
    the separate first pass it replaced."""
    if request.param & _lib.FLAG_COND_THREE_PASS:
        cs = getattr(request.node, 'callspec', None)
        params = cs.params if cs is not None else {}
        about_cond = 'cond' in request.node.name.lower() or bool(params.get('cond')) or 'cond' in str(params.get('golden', ''))
        if not about_cond:
            pytest.skip('the three-pass pin only differs for the conditional posterior')
    monkeypatch.setattr(ops, 'DESC_FLAGS', request.param)

TOL_ELBO = 1e-4


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


# ---------------------------------------------------------------------------
# (1) goldens from the reference
# ---------------------------------------------------------------------------
def test_golden_through_module(golden):
    d = dev()
    model = build_model(golden).to(d)
    golden.response, golden.mask = golden.response.to(d), golden.mask.to(d)
    golden.eps_item, golden.eps_ability = golden.eps_item.to(d), golden.eps_ability.to(d)
    outs, loss = run_reference_pattern(model, golden)
    check_against_golden(model, golden, outs, loss, tol_loss=TOL_ELBO, tol_grad=1e-4, strict=True)


def test_golden_adam_trajectory(golden):
    """3 Adam steps through the HIP path land on the reference's parameters."""
    d = dev()
    model = build_model(golden).to(d)
    golden.response, golden.mask = golden.response.to(d), golden.mask.to(d)
    golden.eps_item, golden.eps_ability = golden.eps_item.to(d), golden.eps_ability.to(d)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    for step in range(3):
        opt.zero_grad()
        _, loss = run_reference_pattern(model, golden)
        loss.backward()
        opt.step()
    for k, v in golden.adam3.items():
        assert (model.state_dict()[k].cpu() - v).abs().max() < 5e-4, k


@pytest.mark.parametrize('rows', ['direct', 'gathered', 'cell-codes'])
def test_golden_adam_trajectory_through_the_fused_trainers(golden, rows):
    """The reference's own recorded Adam steps (tools/gen_golden.py: parameters after 1 and 3 steps of vibo.py:243-268 with
    the case's noise replayed) through the NATIVE train step -- FusedTrainer's kernels for the plain model,
    FusedCondFlowTrainer's (vibo_ctrain_*: table MLP / flow backward + Adam by hand) for --conditional-posterior /
    --n-norm-flows: no autograd, no torch.optim between the goldens and the kernels."""
    from vibo_amd.trainer import FusedTrainer, fused_trainer_covers
    m = golden.meta
    model = build_model(golden)
    if not fused_trainer_covers(model) or m.get('generative_model', 'irt') != 'irt':
        pytest.skip('configuration trains through the module path')
    if m['n_norm_flows'] == 0 and not m['use_kl_divergence']:
        pytest.skip('the fused trainers use the analytic KL regulariser (the CLI default)')
    if rows != 'direct' and m['ability_dim'] > 8:
        pytest.skip('ability_dim 9..16 runs on the wave-per-person kernel: fp32 rows handed over directly')
    d = dev()
    model = model.to(d)
    tr = FusedTrainer(model, lr=5e-3)
    resp, mask = ops.pad_rows(golden.response.to(d), golden.mask.to(d).bool())      # (row strides padded to 4 cells, as the CLI's resident splits)
    eps_i, eps_a = golden.eps_item.to(d), golden.eps_ability.to(d)
    # rows: the golden's minibatch handed over directly, or the way the CLI's resident data path does it -- as a
    # row_index vector into a larger resident matrix (the golden's rows scattered among decoys), or as 1-byte cell codes
    row_index = None
    if rows == 'gathered':
        B, I = golden.response.shape
        g = torch.Generator().manual_seed(B * I)
        big_r = (torch.rand(3 * B + 5, I, generator=g) < 0.5).float()
        big_m = torch.rand(3 * B + 5, I, generator=g) < 0.8
        where = torch.randperm(3 * B + 5, generator=g)[:B]
        big_r[where], big_m[where] = golden.response, golden.mask.bool()
        resp, mask = ops.pad_rows(big_r.to(d), big_m.to(d))
        row_index = where.to(d)
    elif rows == 'cell-codes':
        resp, mask = ops.pack_cell_codes(golden.response.to(d), golden.mask.to(d).bool()), None
    for step in range(3):
        loss = tr.step(resp, mask, beta=m['annealing_factor'], row_index=row_index, eps_item=eps_i, eps_ability=eps_a)
        if step == 0:
            assert rel_err(loss, golden.out['loss']) < TOL_ELBO
            for k, v in golden.adam1.items():
                assert (model.state_dict()[k].cpu() - v).abs().max() < 2e-4, (k, 'after one step')
    for k, v in golden.adam3.items():
        assert (model.state_dict()[k].cpu() - v).abs().max() < 5e-4, k


# ---------------------------------------------------------------------------
# (2) raw kernel outputs vs the CPU analytic oracle
# ---------------------------------------------------------------------------
def random_problem(irt, A, B, I, missing, seed, cond=False, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=missing)
    D = O.item_feat_dim(irt, A)
    table = torch.randn((2, I, 2 * A) if cond else (2, 2 * A), generator=g) * 0.7
    item = torch.randn(I, D, generator=g) * scale
    eps = torch.randn(B, A, generator=g)
    return resp, mask, table, item, eps


def run_kernel(spec, resp, mask, table, item, eps, reg_mode=_lib.REG_KL, mask_dtype=torch.bool,
               row_index=None, want_grad=True, keep_int64=False):
    d = dev()
    r = ops.prepare_response(resp.to(d))
    m, code = ops.prepare_mask(mask.to(d).to(mask_dtype) if mask is not None else None, keep_int64=keep_int64)
    ri = row_index.to(d) if row_index is not None else None
    B = int(ri.numel()) if ri is not None else r.shape[0]
    raw = ops._hip_launch_elbo(spec, r, m, code, ri, table.to(d).contiguous(), item.to(d).contiguous(),
                               eps.to(d).contiguous(), None, reg_mode, want_grad, B)
    torch.cuda.synchronize()
    return raw


# Gradient tolerances (fractions of the tensor's max-abs against the fp64 analytic oracle).  SURVEY.md section 8c asks for <= 1e-4;
# TOL_GRAD is that bound and the default everywhere.  Named wider bands exist only where the fp64 oracle cannot arbitrate to 1e-4,
# each with the reason and the maximum measured on the GPU (VIBO_TOL_RECORD=path appends every observed error to a JSON-lines
# file; gpurun_out/r6_tolerances.jsonl is the record these numbers come from):
TOL_GRAD = 1e-4
_OBSERVED = []


def _check(kind, err, tol):
    if os.environ.get('VIBO_TOL_RECORD'):
        _OBSERVED.append((kind, float(err), float(tol), os.environ.get('PYTEST_CURRENT_TEST', '')))
        with open(os.environ['VIBO_TOL_RECORD'], 'a') as f:
            f.write(json.dumps({'kind': kind, 'err': float(err), 'tol': float(tol), 'test': os.environ.get('PYTEST_CURRENT_TEST', '')}) + '\n')
        if os.environ.get('VIBO_TOL_RECORD_ONLY'):
            return
    assert err < tol, (kind, err, tol)


def compare_raw(raw, ref, item_shape, want_grad=True, tol=TOL_GRAD):
    sc = raw.scalars.cpu()
    assert rel_err(sc[_lib.S_LL], ref['ll']) < 2e-5
    assert abs(float(sc[_lib.S_REG]) - float(ref['reg'])) < 2e-5 * max(1.0, abs(float(ref['reg'])))
    assert abs(float(sc[_lib.S_KL]) - float(ref['kl_ability'])) < 2e-5 * max(1.0, abs(float(ref['kl_ability'])))
    assert abs(float(sc[_lib.S_LOGQ0]) - float(ref['logq0'])) < 2e-5 * max(1.0, abs(float(ref['logq0'])))
    assert abs(float(sc[_lib.S_LOGP]) - float(ref['logp'])) < 2e-5 * max(1.0, abs(float(ref['logp'])))
    for k, t in (('ability_mu', raw.ability_mu), ('ability_logvar', raw.ability_logvar), ('ability', raw.ability)):
        assert (t.cpu() - ref[k].float()).abs().max() < 2e-5 * max(1.0, float(ref[k].abs().max())), k
    if want_grad:
        for s in range(2):
            scale = float(ref['g_table'][s].abs().max())
            if scale > 0:
                _check(f'g_table[{s}]', rel_err(raw.grad_table(s).cpu(), ref['g_table'][s]), tol)
            else:
                assert float(raw.grad_table(s).abs().max()) < 1e-6
        _check('g_item', rel_err(raw.grad_item(item_shape).cpu(), ref['g_item']), tol)


SHAPES = [
    # irt, A, B, I, missing
    (2, 1, 64, 1000, 0.0),
    (2, 1, 200, 1000, 0.2),
    (2, 8, 130, 1000, 0.1),
    (2, 8, 64, 1024, 0.0),
    (2, 4, 70, 600, 0.3),       # 3 waves per row, last wave partly empty
    (2, 2, 33, 1016, 0.1),
    (2, 2, 300, 304, 0.1),      # 2 waves per row
    (2, 8, 150, 144, 0.2),      # 1 wave per row
    (2, 3, 257, 100, 0.2),      # A padded 3 -> 4
    (2, 5, 65, 512, 0.0),       # A padded 5 -> 8
    (2, 1, 31, 95, 0.2),        # ragged contiguous rows (I % 4 != 0): tiled kernel, scalar load path
    (2, 7, 100, 130, 0.1),
    (2, 1, 5, 1, 0.0),          # single item (tiled kernel)
    (2, 2, 1, 7, 0.0),          # single person
    (1, 1, 100, 1000, 0.1),
    (1, 4, 77, 333, 0.2),
    (3, 1, 100, 1000, 0.1),
    (3, 8, 90, 640, 0.0),
    (3, 2, 50, 95, 0.3),
    (3, 6, 64, 1003, 0.1),
    # more row-split shapes: 1..4 waves per row, batch tails
    (2, 3, 257, 256, 0.2),
    (2, 6, 41, 260, 0.1),
    (2, 8, 1001, 768, 0.1),
    (1, 8, 64, 1000, 0.1),
    (1, 4, 77, 332, 0.2),
    (2, 4, 9, 1024, 0.5),
    (2, 7, 3, 196, 0.0),
]


@pytest.mark.parametrize('irt,A,B,I,missing', SHAPES)
@pytest.mark.parametrize('drop', [False, True])
def test_raw_kernel_vs_oracle(irt, A, B, I, missing, drop):
    spec = ElboSpec(irt_model=irt, ability_dim=A, drop_missing=drop)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, missing, seed=B * 131 + I + A)
    if drop and missing > 0:
        mask[:, 0] = 1          # --drop-missing needs >= 1 observed cell per person (else 0/0, as in the reference)
        resp[:, 0] = resp[:, 0].clamp(min=0)
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(),
                           irt_model=irt, ability_dim=A, replace_missing_with_prior=not drop, mode='kl')
    raw = run_kernel(spec, resp, mask, table, item, eps)
    compare_raw(raw, ref, (I, spec.item_dim))


@pytest.mark.parametrize('mask_dtype,keep_int64', [(torch.bool, False), (torch.int64, False), (torch.int64, True),
                                                   (torch.uint8, False), (None, False)])
@pytest.mark.parametrize('I,A', [(1000, 2), (95, 2), (600, 5)])
def test_mask_dtypes(mask_dtype, keep_int64, I, A):
    """bool / uint8 in place; int64 narrowed once by ops.prepare_mask, or handed to the library as VIBO_MASK_I64
    (wave-per-row kernel for A <= 2, tiled kernel otherwise)."""
    irt, B = 2, 150
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.0 if mask_dtype is None else 0.25, seed=7)
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(),
                           irt_model=irt, ability_dim=A, mode='kl')
    raw = run_kernel(spec, resp, None if mask_dtype is None else mask, table, item, eps,
                     mask_dtype=mask_dtype or torch.bool, keep_int64=keep_int64)
    compare_raw(raw, ref, (I, spec.item_dim))


def test_int64_mask_is_narrowed_once_per_tensor():
    d = dev()
    mask = (torch.rand(64, 100, device=d) > 0.2).long()
    a, code = ops.prepare_mask(mask)
    b, _ = ops.prepare_mask(mask)
    assert code == _lib.MASK_U8 and a.dtype == torch.uint8 and a.data_ptr() == b.data_ptr()      # cached
    mask[0, 0] = 1 - mask[0, 0]                                                                    # in-place edit
    c, _ = ops.prepare_mask(mask)
    assert c.data_ptr() != a.data_ptr() and int(c[0, 0]) == int(mask[0, 0])
    assert ops.prepare_mask(mask, keep_int64=True)[1] == _lib.MASK_I64


GENERAL_SHAPES = [
    # irt, A, B, I, missing, cond, n_flows      (conditional / flows / > 1024 items: row-split paths where the rows
    # are 16-byte chunkable, wave-per-person kernel otherwise)
    (2, 1, 70, 1500, 0.2, False, 0),
    (2, 8, 33, 2048, 0.1, False, 0),
    (3, 2, 40, 1203, 0.1, False, 0),
    (2, 1, 100, 200, 0.2, True, 0),
    (2, 8, 50, 130, 0.1, True, 0),
    (3, 2, 64, 95, 0.3, True, 0),
    (1, 3, 77, 333, 0.0, True, 0),
    (2, 1, 100, 100, 0.2, False, 4),
    (2, 3, 65, 257, 0.1, False, 2),
    (3, 1, 80, 1000, 0.1, True, 4),
    (1, 2, 30, 64, 0.0, False, 1),
    # planar flows inside the row-split kernel (192 <= I <= 1024, I % 4 == 0)
    (2, 1, 100, 1000, 0.2, False, 4),
    (2, 8, 77, 1000, 0.1, False, 2),
    (3, 2, 50, 256, 0.1, False, 3),
    (1, 4, 33, 600, 0.0, False, 8),
    (3, 5, 21, 1024, 0.3, False, 1),
    (2, 3, 130, 196, 0.1, False, 4),
    # more than 1024 items, unconditional: row-count pass + one row-split launch per 1024-item panel
    (2, 1, 100, 2500, 0.2, False, 4),
    (1, 3, 50, 1028, 0.1, False, 0),
    (3, 8, 30, 3000, 0.1, False, 0),
    (2, 2, 9, 10000, 0.3, False, 2),
    (2, 5, 200, 4096, 0.0, False, 0),
    # conditional posterior through cond_pre / row-split / cond_post (ability_dim <= 4, 192 <= I, I % 4 == 0)
    (2, 2, 130, 1000, 0.2, True, 0),
    (2, 4, 33, 600, 0.1, True, 2),
    (3, 1, 20, 2500, 0.1, True, 4),
    (1, 1, 50, 192, 0.0, True, 0),
    (2, 3, 41, 1024, 0.3, True, 0),
    (3, 1, 9, 10000, 0.2, True, 4),
    (2, 4, 15, 64, 0.1, True, 1),          # one-wave workgroups with 9 PoE sums per row (found by tools/fuzz_parity.py)
    (2, 3, 64, 8, 0.0, True, 0),
    # ability_dim 9..16 (vibo.py:36-37 takes any int): the wave-per-person kernel's wide instantiation, every configuration
    (2, 9, 70, 1000, 0.2, False, 0),
    (2, 12, 33, 130, 0.1, True, 0),
    (3, 16, 40, 95, 0.1, False, 2),
    (1, 10, 50, 333, 0.0, True, 3),
    (2, 16, 21, 1500, 0.3, True, 8),
]


@pytest.mark.parametrize('irt,A,B,I,missing,cond,n_flows', GENERAL_SHAPES)
@pytest.mark.parametrize('drop', [False, True])
def test_general_kernel_vs_oracle(irt, A, B, I, missing, cond, n_flows, drop):
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=cond, drop_missing=drop, n_flows=n_flows)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, missing, seed=B * 7 + I + A, cond=cond)
    if drop and missing > 0:
        mask[:, 0] = 1
        resp[:, 0] = resp[:, 0].clamp(min=0)
    g = torch.Generator().manual_seed(I)
    flow = None
    flows = None
    if n_flows:
        flow = torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5
        flows = [(f[:A].double(), f[A:2 * A].double(), f[2 * A:].double()) for f in flow]
    mode = 'sampled' if n_flows else 'kl'
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(), irt_model=irt,
                           ability_dim=A, conditional_posterior=cond, replace_missing_with_prior=not drop,
                           mode=mode, flow_uhat_w_b=flows)
    d = dev()
    r = ops.prepare_response(resp.to(d))
    m, code = ops.prepare_mask(mask.bool().to(d))
    raw = ops._hip_launch_elbo(spec, r, m, code, None, table.to(d).contiguous(), item.to(d).contiguous(),
                               eps.to(d).contiguous(), flow.to(d).contiguous() if flow is not None else None,
                               _lib.REG_SAMPLED if n_flows else _lib.REG_KL, True, B)
    torch.cuda.synchronize()
    compare_raw(raw, ref, (I, spec.item_dim))
    if n_flows:
        assert (raw.ability_k.cpu() - ref['ability_k'].float()).abs().max() < 5e-5
        assert (raw.ability_ladj.cpu() - ref['ladj'].float()).abs().max() < 5e-5
        assert abs(float(raw.scalars[_lib.S_LADJ]) - float(ref['ladj_sum'])) < 1e-4 * max(1.0, abs(float(ref['ladj_sum'])))
        for s_ in range(2):
            gref = torch.cat([torch.cat(gf) for gf in ref['g_flow'][s_]]).float()
            _check(f'g_flow[{s_}]', rel_err(raw.grad_flow(s_).cpu(), gref), TOL_GRAD)


def test_sampled_regulariser_mode():
    irt, A, B, I = 2, 3, 100, 200
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.2, seed=11)
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(),
                           irt_model=irt, ability_dim=A, mode='sampled')
    raw = run_kernel(spec, resp, mask, table, item, eps, reg_mode=_lib.REG_SAMPLED)
    compare_raw(raw, ref, (I, spec.item_dim))


def test_row_index_gather_and_strided_rows():
    irt, A, P, I = 2, 2, 500, 1000
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    resp, mask, table, item, _ = random_problem(irt, A, P, I, 0.2, seed=5)
    idx = torch.randperm(P, generator=torch.Generator().manual_seed(1))[:130]
    eps = torch.randn(130, A, generator=torch.Generator().manual_seed(2))
    ref = T.fused_elbo_ref(table.double(), item.double(), resp[idx].double(), mask[idx], eps.double(),
                           irt_model=irt, ability_dim=A, mode='kl')
    raw = run_kernel(spec, resp, mask, table, item, eps, row_index=idx)
    compare_raw(raw, ref, (I, spec.item_dim))


@pytest.mark.parametrize('irt,A,B,I,cond,n_flows', [(3, 4, 129, 777, False, 0), (2, 8, 200, 1000, False, 0),
                                                   (2, 1, 77, 1000, False, 4), (2, 2, 65, 2052, False, 0),
                                                   (2, 1, 90, 600, True, 0), (3, 1, 33, 2500, True, 4)])
def test_forward_only_matches_forward_of_train(irt, A, B, I, cond, n_flows):
    """want_grad = 0 (eval / log_marginal): same heads and posteriors as the training launch, on every path
    (tiled, row-split, panels, conditional)."""
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=cond, n_flows=n_flows)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.1, seed=3, cond=cond)
    g = torch.Generator().manual_seed(I)
    flow = torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5 if n_flows else None
    d = dev()
    r = ops.prepare_response(resp.to(d))
    m, code = ops.prepare_mask(mask.bool().to(d))
    outs = []
    for want_grad in (True, False):
        outs.append(ops._hip_launch_elbo(spec, r, m, code, None, table.to(d).contiguous(), item.to(d).contiguous(),
                                         eps.to(d).contiguous(), flow.to(d).contiguous() if flow is not None else None,
                                         _lib.REG_SAMPLED if n_flows else _lib.REG_KL, want_grad, B))
    torch.cuda.synchronize()
    a, b = outs
    assert rel_err(b.scalars[:7].cpu(), a.scalars[:7].cpu()) < 1e-6
    # two template instantiations of the same source: the compiler may contract a*b+c differently, so last-bit equal
    assert float((a.ability_mu - b.ability_mu).abs().max()) < 1e-6 and float((a.ability - b.ability).abs().max()) < 1e-6
    if n_flows:
        assert float((a.ability_k - b.ability_k).abs().max()) < 2e-6


def test_all_missing_rows_and_saturated_logits():
    """Rows with a single observed cell / none observed (prior mode), and item
    parameters large enough to drive logits past the Bernoulli clamp."""
    irt, A, B, I = 2, 1, 64, 400
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.3, seed=9, scale=12.0)
    mask[0] = 0
    resp[0] = -1
    mask[1] = 0
    mask[1, 17] = 1
    resp[1] = -1
    resp[1, 17] = 1
    ref = T.fused_elbo_ref(table, item, resp, mask, eps, irt_model=irt, ability_dim=A, mode='kl',
                           exact_saturation=True)
    assert float((ref['logit'].abs() > 17).float().mean()) > 0.02      # the clamp really is exercised
    raw = run_kernel(spec, resp, mask, table, item, eps)
    compare_raw(raw, ref, (I, spec.item_dim))


@pytest.mark.parametrize('irt,A,scale', [(2, 1, 1e-4), (2, 8, 1e-4), (3, 2, 1e-4), (1, 4, 1e-4),
                                         (2, 1, 1e3), (2, 8, 1e3), (1, 4, 1e3), (2, 1, 1e5), (2, 8, 1e5), (2, 1, 3e6), (2, 8, 3e6)])
def test_item_scales_far_outside_the_f16_range(irt, A, scale):
    """The matrix kernel's contractions run on f16 hi/lo pieces (vibo_msplit_kernel.hpp): item parameters far below and far
    above the f16 range (max 65 504) must still give the fp32 reference's numbers -- the kernel balances theta against the
    discriminations and rescales the difficulties by powers of two per launch (exact).  Discriminations / difficulties of
    1e-4 ... 3e6: beyond ~1e3 most logits are past the Bernoulli clamp (log-lik capped, likelihood gradients exactly zero),
    which the oracle reproduces (exact_saturation); at 3e6 all of them are.  The few cells whose logit -a.theta + b cancels to
    O(1) carry gradients of O(scale) computed from an fp32 logit with an absolute error of ~scale x 6e-8 -- in the
    reference's own arithmetic as much as here -- hence the tolerance that grows with the scale (an f16 piece saturating at
    65 504 would be an O(1) error).  3PL only at the small scale: its clamp acts on p = guess + (1 - guess) sigmoid(l) and
    flips with the rounding of that sum.  Runs on both row-split kernels (fixture)."""
    B, I = 96, 384
    if scale > 1e5 and (ops.DESC_FLAGS & _lib.FLAG_KERNEL_VALU):
        # round 1's VALU kernel clamps the logit BEFORE the exponential: a cell on the likely side of the clamp keeps
        # d ll/d l = -+2^-23 where the reference has an exact 0 -- 1.2e-7 per cell, visible only when multiplied by
        # discriminations of millions.  Known, documented; the matrix kernel has no such term.
        pytest.skip('VALU kernel: likely-side clamp residue of 2^-23 per cell x |a| = 3e6')
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.2, seed=31 + A)
    if irt == 1:
        item *= scale
    else:
        item[:, :A + 1] *= scale               # discriminations and difficulties (the 3PL guess logit stays O(1))
    ref = T.fused_elbo_ref(table, item, resp, mask, eps, irt_model=irt, ability_dim=A, mode='kl', exact_saturation=True)
    raw = run_kernel(spec, resp, mask, table, item, eps)
    assert torch.isfinite(raw.flat).all()
    compare_raw(raw, ref, (I, spec.item_dim), tol=max(TOL_GRAD, 1e-6 * scale))      # (measured: 2.1e-4 at scale 1e3, 1.5e-5 at 1e5, 8.5e-6 at 1e-4)


def test_operands_beyond_the_rescaling_range_fail_loudly():
    """A difficulty beyond 2^30 (or an ability sample beyond the rescaled f16 range) cannot be represented by the matrix
    kernel's f16 pieces: the result is NaN, never a silently wrong number; the fp32 VALU kernel (VIBO_FLAG_KERNEL_VALU)
    still evaluates it."""
    irt, A, B, I = 2, 2, 64, 256
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.1, seed=5)
    item[3, A] = 3e9
    raw = run_kernel(spec, resp, mask, table, item, eps)
    if ops.DESC_FLAGS & _lib.FLAG_KERNEL_VALU:
        ref = T.fused_elbo_ref(table, item, resp, mask, eps, irt_model=irt, ability_dim=A, mode='kl', exact_saturation=True)
        assert rel_err(raw.scalars.cpu()[_lib.S_LL], ref['ll']) < 2e-5
    else:
        assert torch.isnan(raw.scalars[_lib.S_LL])


def test_saturation_golden_through_kernel():
    """One person (theta = 0 => logit = b_i), items with difficulties on the
    reference's saturation probe grid: per-item gradients must be exactly zero where
    the reference's are."""
    import numpy as np
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'saturation.npz'))
    logit = torch.from_numpy(z["logit"])[:2000:2]
    I = logit.numel()
    spec = ElboSpec(irt_model=2, ability_dim=1)
    table = torch.zeros(2, 2)
    item = torch.stack([torch.ones(I), logit], dim=1)      # a_i = 1, b_i = logit; theta forced to 0 below
    for x in (0, 1):
        resp = torch.full((1, I), float(x))
        mask = torch.ones(1, I, dtype=torch.bool)
        # table = 0 => tau = 1, mu = 0; eps = 0 => theta = 0 exactly
        raw = run_kernel(spec, resp, mask, table, item, torch.zeros(1, 1))
        g_b = raw.grad_item((I, 2))[:, 1].cpu()
        ref_g = torch.from_numpy(z[f'dll_dlogit_x{x}'])[:2000:2]
        assert torch.equal(g_b == 0, ref_g == 0)
        assert (g_b - ref_g).abs().max() < 2e-6
        # value: exact (fp64) clamped log-likelihood to 1e-5; the reference's own fp32 sum sits up to
        # ~3e-4 away from exact arithmetic on this adversarial grid (its log(1-p) loses bits for
        # |logit| > 10), so it is only a loose bound here
        lc = logit.double().clamp(-T.LOGIT_LO, T.LOGIT_LO)
        ll_exact = float((x * lc - lc.clamp(min=0) - torch.log1p(torch.exp(-lc.abs()))).sum())
        ll_ref = float(torch.from_numpy(z[f'll_x{x}'])[:2000:2].double().sum())
        assert abs(float(raw.scalars[_lib.S_LL]) - ll_exact) < 1e-5 * abs(ll_exact)
        assert abs(float(raw.scalars[_lib.S_LL]) - ll_ref) < 5e-4 * abs(ll_ref)


def test_saturation_3pl_golden_through_kernel():
    """3PL in and around the Bernoulli probability clamp, against the reference (tests/golden/saturation_3pl.npz: p = g +
    (1 - g) sigmoid(l) through irt_model_3pl and masked_bernoulli_log_pdf).  One person with theta = 0, items with unit
    discrimination, difficulty = the probe logit, three guess logits.  Where the reference's rounded p sits on the clamp
    threshold 1 - eps32 (1 to 3 ulp below 1: p is constant over ~0.5 logit there, the decision is a coin flip of its last bit)
    a cell's O(1) gradient legitimately flips; everywhere else -- below the band, and where p has rounded to 1 -- value and
    both item gradients must be the reference's, zeros included."""
    import os
    import numpy as np
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'saturation_3pl.npz'))
    logit = torch.from_numpy(z['logit'])
    I = logit.numel()
    I4 = (I + 3) // 4 * 4
    spec = ElboSpec(irt_model=3, ability_dim=1)
    table = torch.zeros(2, 2)
    eps32 = float(np.finfo(np.float32).eps)
    for gi, gl in enumerate(z['guess_logit']):
        item = torch.zeros(I4, 3)
        item[:I, 0], item[:I, 1], item[:, 2] = 1.0, logit, float(gl)
        p_ref = torch.from_numpy(z[f'p_g{gi}'])
        # 1 - p in {1, 2, 3} ulp: a 1-ulp difference in p decides between "clamped" (p > 1 - eps32 = 1 - 2 ulp) and not
        d1 = 1.0 - p_ref.double()
        on_edge = (d1 >= 0.5 * 2.0 ** -24) & (d1 <= 3.5 * 2.0 ** -24)
        assert int(on_edge.sum()) < 0.25 * I          # (half the grid is a dense sweep of the band itself)
        for x in (0, 1):
            resp = torch.full((1, I4), float(x))
            mask = torch.zeros(1, I4, dtype=torch.bool)
            mask[:, :I] = True
            raw = run_kernel(spec, resp, mask, table, item, torch.zeros(1, 1))
            g = raw.grad_item((I4, 3)).cpu()[:I]
            for col, key in ((1, 'dll_db'), (2, 'dll_dguess')):
                ref = torch.from_numpy(z[f'{key}_g{gi}_x{x}'])
                ok = ~on_edge
                err = (g[:, col] - ref).abs()
                # the reference forms 1 - p in fp32: its own gradient carries a relative error of ~ulp(1) / (1 - p) there
                # (3 % at logit 13, where 1 - p is 40 ulp); the kernel works from P(wrong) = (1 - g) sigmoid(-l) directly
                rel = 2e-4 + 2.0 * 2.0 ** -24 / d1.clamp_min(2.0 ** -24).float()
                assert float((err[ok] - rel[ok] * ref[ok].abs()).max()) < 3e-6, (gi, x, key)
                # exact zeros of the reference (p clamped): zero here as well, away from the threshold plateau
                zero = ok & (ref == 0)
                assert float(g[zero, col].abs().max() if zero.any() else 0.0) < 1e-6, (gi, x, key)
            ll_ref = torch.from_numpy(z[f'll_g{gi}_x{x}']).double()
            # (edge cells can differ by the clamp's jump in the VALUE only through p's last bit: tiny)
            assert abs(float(raw.scalars[_lib.S_LL]) - float(ll_ref.sum())) < 5e-4 * abs(float(ll_ref.sum()))


# ---------------------------------------------------------------------------
# encode / decode entry points
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('cond', [False, True])
@pytest.mark.parametrize('drop', [False, True])
@pytest.mark.parametrize('A,B,I', [(3, 101, 333), (1, 100, 1000), (8, 77, 600), (4, 33, 2500), (2, 9, 96), (5, 50, 1028)])
def test_encode_kernel(cond, drop, A, B, I):
    """model.encode's kernel: wave-per-person (ragged rows, conditional with A > 4) and the row-statistics fast path
    (row_count / cond_pre + per-person finish), with and without row gather."""
    irt = 2
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=cond, drop_missing=drop)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.2, seed=21, cond=cond)
    mask[:, 0] = 1
    resp[:, 0] = resp[:, 0].clamp(min=0)
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(), irt_model=irt,
                           ability_dim=A, conditional_posterior=cond, replace_missing_with_prior=not drop,
                           mode='kl', want_grad=False)
    d = dev()
    mu, lv = ops.encode_posterior(spec, table.to(d), resp.to(d), mask.bool().to(d))
    assert (mu.cpu() - ref['ability_mu'].float()).abs().max() < 2e-5 * max(1.0, float(ref['ability_mu'].abs().max()))
    assert (lv.cpu() - ref['ability_logvar'].float()).abs().max() < 2e-5 * max(1.0, float(ref['ability_logvar'].abs().max()))
    rows = torch.randperm(B, generator=torch.Generator().manual_seed(B))[:max(1, B // 2)]
    mu2, lv2 = ops.encode_posterior(spec, table.to(d), resp.to(d), mask.bool().to(d), row_index=rows.to(d))
    assert (mu2.cpu() - mu.cpu()[rows]).abs().max() < 1e-6 and (lv2.cpu() - lv.cpu()[rows]).abs().max() < 1e-6


@pytest.mark.parametrize('irt', [1, 2, 3])
def test_decode_kernel(irt):
    A, B, I = 3, 70, 211
    g = torch.Generator().manual_seed(irt)
    ability = torch.randn(B, A, generator=g)
    item = torch.randn(I, O.item_feat_dim(irt, A), generator=g)
    d = dev()
    out = ops.decode_probs(ElboSpec(irt_model=irt, ability_dim=A), ability.to(d), item.to(d))
    assert (out.cpu() - O.irt_link(irt, ability, item)).abs().max() < 2e-6


# ---------------------------------------------------------------------------
# (3) properties at BASELINE.json sizes
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('A,P', [(1, 100_000), (8, 100_000), (8, 1_000_000)])
def test_full_size_shard_additivity_permutation_determinism(A, P):
    """100k x 1k 2PL (BASELINE configs[1] shape; [2]'s ability_dim) and configs[2] at its FULL size, 1M x 1k at
    ability_dim 8 (the benchmark's matrix): the ELBO heads
    and gradients of the whole batch equal the sum over two person shards (what
    the 8-GPU person sharding relies on), are invariant to permuting persons,
    and are bitwise reproducible."""
    irt, I = 2, 1000
    d = dev()
    g = torch.Generator(device=d).manual_seed(1234)
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    theta = torch.randn(P, A, device=d, generator=g)
    item_true = torch.randn(I, A + 1, device=d, generator=g)
    resp = torch.empty(P, I, device=d)
    mask = torch.empty(P, I, dtype=torch.bool, device=d)
    for s0 in range(0, P, 100_000):         # (in slices: the 1M-person case would hold three more 4 GB temporaries)
        sl = slice(s0, min(P, s0 + 100_000))
        probs = torch.sigmoid(-(theta[sl] @ item_true[:, :A].t()) + item_true[:, A])
        resp[sl] = torch.bernoulli(probs, generator=g)
        mask[sl] = torch.rand(probs.shape, device=d, generator=g) > 0.1
    del probs
    table = (torch.randn(2, 2 * A, device=d, generator=g) * 0.5).contiguous()
    item = torch.randn(I, A + 1, device=d, generator=g)
    eps = torch.randn(P, A, device=d, generator=g)

    def run(rows):
        r, m = resp[rows].contiguous(), mask[rows].contiguous()
        mm, code = ops.prepare_mask(m)
        raw = ops._hip_launch_elbo(spec, r, mm, code, None, table, item, eps[rows].contiguous(), None,
                                   _lib.REG_KL, True, r.shape[0])
        return raw

    allrows = torch.arange(P, device=d)
    full = run(allrows)
    again = run(allrows)
    assert torch.equal(full.flat, again.flat), 'fused kernel must be bitwise deterministic'
    h = P // 2 + 37
    a, b = run(allrows[:h]), run(allrows[h:])
    summed = a.flat + b.flat
    assert rel_err(summed[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(summed[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    perm = torch.randperm(P, device=d, generator=g)
    pf = run(perm)
    assert rel_err(pf.flat[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(pf.flat[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    assert (pf.ability_mu - full.ability_mu[perm]).abs().max() < 1e-6
    # the same permutation through the in-kernel row gather (row_index) is the same call bit for bit
    mm, code = ops.prepare_mask(mask)
    pg = ops._hip_launch_elbo(spec, resp, mm, code, perm, table, item, eps[perm].contiguous(), None, _lib.REG_KL, True, P)
    assert torch.equal(pg.flat, pf.flat) and torch.equal(pg.ability_mu, pf.ability_mu)

    # and the full-size ELBO agrees with the CPU oracle on a 512-person slice
    sl = allrows[:512]
    ref = T.fused_elbo_ref(table.cpu().double(), item.cpu().double(), resp[sl].cpu().double(), mask[sl].cpu(),
                           eps[sl].cpu().double(), irt_model=irt, ability_dim=A, mode='kl')
    compare_raw(run(sl), ref, (I, A + 1))


def _device_problem(irt, A, P, I, missing, seed, cond):
    """A simulated response matrix built on the GPU in person slices (the full-size cases do not fit the host-side helper's
    temporaries): responses from the model's own link (models.py:729-766), `missing` of the cells unobserved."""
    d = dev()
    g = torch.Generator(device=d).manual_seed(seed)
    D = O.item_feat_dim(irt, A)
    theta = torch.randn(P, A, device=d, generator=g)
    item_true = torch.randn(I, D, device=d, generator=g)
    resp = torch.empty(P, I, device=d)
    mask = torch.empty(P, I, dtype=torch.bool, device=d)
    step = max(1, 100_000_000 // I)
    for s0 in range(0, P, step):
        sl = slice(s0, min(P, s0 + step))
        if irt == 1:
            logit = theta[sl].sum(1, keepdim=True) + item_true[:, 0]
        else:
            logit = -(theta[sl] @ item_true[:, :A].t()) + item_true[:, A]
        probs = torch.sigmoid(logit)
        if irt == 3:
            gs = torch.sigmoid(item_true[:, A + 1])
            probs = gs + (1 - gs) * probs
        resp[sl] = torch.bernoulli(probs, generator=g)
        mask[sl] = torch.rand(probs.shape, device=d, generator=g) >= missing
        del logit, probs
    table = (torch.randn((2, I, 2 * A) if cond else (2, 2 * A), device=d, generator=g) * 0.5).contiguous()
    item = torch.randn(I, D, device=d, generator=g)
    eps = torch.randn(P, A, device=d, generator=g)
    return resp, mask, table, item, eps


@pytest.mark.parametrize('P', [100_000, 1_000_000], ids=['100k-persons', 'full-size-1M-persons'])
@pytest.mark.parametrize('codes', [False, True], ids=['fp32-rows', 'cell-codes'])
def test_config5_pipeline_at_the_planner_large_call_size(codes, P):
    """BASELINE configs[4]'s row shape at a size the planner's large-call paths see (100 000 persons x 10 000 items, 3PL,
    ability_dim 1, --conditional-posterior, 4 planar flows, 20 % missing; models.py:695-710,758-765, flows.py:21-66): the
    count-and-emit pass over all ten 1024-item panels, the panel sums, the matrix kernel's one-launch panel mode with the flow and
    conditional hooks, and the table-gradient pass -- bitwise reproducible, additive over two person shards (what the 8-GPU
    sharding relies on), the same through an in-kernel row gather, and equal to the CPU oracle on a 48-person slice.
    (The 9-person GENERAL_SHAPES case of this configuration never leaves the small-call kernels.)
    `full-size-1M-persons` = BASELINE configs[4] LITERALLY: 1 000 000 x 10 000 = 1e10 cells (40 GB of fp32 responses + 10 GB of mask
    bytes, or 10 GB of cell codes, resident on the one GPU) -- the first shape whose cell index does not fit 32 bits."""
    irt, A, I, n_flows = 3, 1, 10_000, 4
    d = dev()
    if P * I > 2_000_000_000 and torch.cuda.mem_get_info(d)[0] < 150 * 2 ** 30:
        pytest.skip('the full-size case needs ~150 GB of free HBM')
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=True, n_flows=n_flows)
    resp, mask, table, item, eps = _device_problem(irt, A, P, I, 0.2, seed=55, cond=True)
    g = torch.Generator().manual_seed(5)
    flow = (torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5).to(d).contiguous()
    if codes:
        cells = ops.pack_cell_codes(resp, mask).codes

    def run(rows=None, row_index=None):
        e = eps if rows is None else eps[rows].contiguous()
        if row_index is not None:
            e = eps[row_index].contiguous()
        B = P if rows is None and row_index is None else int((rows if rows is not None else row_index).numel())
        if codes:
            c = cells if rows is None else cells[rows].contiguous()
            out = ops._hip_launch_elbo(spec, c, c, _lib.MASK_CODES, row_index, table, item, e, flow, _lib.REG_SAMPLED, True, B)
        else:
            r = resp if rows is None else resp[rows].contiguous()
            m = mask if rows is None else mask[rows].contiguous()
            mm, code = ops.prepare_mask(m)
            out = ops._hip_launch_elbo(spec, r, mm, code, row_index, table, item, e, flow, _lib.REG_SAMPLED, True, B)
        torch.cuda.synchronize()
        return out

    full, again = run(), run()
    assert torch.equal(full.flat, again.flat) and torch.equal(full.grad_table(0), again.grad_table(0)), 'must be bitwise deterministic'
    assert torch.equal(full.ability_mu, again.ability_mu) and torch.equal(full.ability_k, again.ability_k)
    allrows = torch.arange(P, device=d)
    h = P // 2 + 37
    a, b = run(allrows[:h]), run(allrows[h:])
    summed = a.flat + b.flat
    assert rel_err(summed[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(summed[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    for s_ in range(2):
        assert rel_err((a.grad_table(s_) + b.grad_table(s_)).cpu(), full.grad_table(s_).cpu()) < 1e-4
    assert (torch.cat([a.ability_mu, b.ability_mu]) - full.ability_mu).abs().max() < 1e-6
    # a person permutation through the in-kernel gather: the same sums, the persons' posteriors permuted
    perm = torch.randperm(P, device=d, generator=torch.Generator(device=d).manual_seed(9))
    pg = run(row_index=perm)
    assert rel_err(pg.flat[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(pg.flat[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    assert (pg.ability_mu - full.ability_mu[perm]).abs().max() < 1e-6
    # the CPU oracle on a slice (its own small-call kernels) and the same persons inside the large call
    n = 48
    flows = [(f[:A].double().cpu(), f[A:2 * A].double().cpu(), f[2 * A:].double().cpu()) for f in flow]
    ref = T.fused_elbo_ref(table.cpu().double(), item.cpu().double(), resp[:n].cpu().double(), mask[:n].cpu(),
                           eps[:n].cpu().double(), irt_model=irt, ability_dim=A, conditional_posterior=True, mode='sampled',
                           flow_uhat_w_b=flows)
    compare_raw(run(allrows[:n]), ref, (I, spec.item_dim))
    for k, t in (('ability_mu', full.ability_mu), ('ability_logvar', full.ability_logvar), ('ability', full.ability),
                 ('ability_k', full.ability_k)):
        assert (t[:n].cpu() - ref[k].float()).abs().max() < 2e-5 * max(1.0, float(ref[k].abs().max())), k
    assert (full.ability_ladj[:n].cpu() - ref['ladj'].float()).abs().max() < 5e-5


@pytest.mark.parametrize('irt,P,I', [(2, 1_000_000, 1000), (3, 200_003, 998), (1, 70_000, 516)], ids=['2pl-1Mx1000', '3pl-200003x998', '1pl-70000x516'])
def test_conditional_posterior_one_dim_at_full_size(irt, P, I, monkeypatch):
    """--conditional-posterior at ability_dim 1 on fp32 rows at the sizes the matrix kernel serves (models.py:664-710): under the
    first pin the kernel forms the experts' sums itself and emits the rows' cell codes for the table-gradient pass (its XM == 3; 8 waves
    at 1000 / 998 items, 5 at 516, a ragged last chunk at 998), under the others the separate first pass runs.  Bitwise reproducible,
    additive over two person shards, invariant under a person permutation (in-kernel gather), equal to the three passes within fp32
    rounding, and equal to the fp64 oracle on a slice."""
    A = 1
    d = dev()
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=True)
    resp, mask, table, item, eps = _device_problem(irt, A, P, I, 0.2, seed=66, cond=True)
    rp, mp = ops.pad_rows(resp, mask)

    def run(rows=None, row_index=None, grad=True):
        r = rp if rows is None else rp[rows]
        m = mp if rows is None else mp[rows]
        e = eps if rows is None else eps[rows].contiguous()
        if row_index is not None:
            e = eps[row_index].contiguous()
        r2, m2, code = ops.prepare_rows(r, m)
        B = int(row_index.numel()) if row_index is not None else r.shape[0]
        out = ops._hip_launch_elbo(spec, r2, m2, code, row_index, table, item, e, None, _lib.REG_KL, grad, B)
        torch.cuda.synchronize()
        return out

    full, again = run(), run()
    assert torch.equal(full.flat, again.flat) and torch.equal(full.ability_mu, again.ability_mu), 'must be bitwise deterministic'
    h = P // 2 + 37
    a, b = run(slice(0, h)), run(slice(h, P))
    summed = a.flat + b.flat
    assert rel_err(summed[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(summed[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    assert (torch.cat([a.ability_mu, b.ability_mu]) - full.ability_mu).abs().max() < 1e-6
    perm = torch.randperm(P, device=d, generator=torch.Generator(device=d).manual_seed(9))
    pg = run(row_index=perm)
    assert rel_err(pg.flat[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(pg.flat[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    assert (pg.ability_mu - full.ability_mu[perm]).abs().max() < 1e-6
    fwd = run(grad=False)                    # forward only: no codes leave the kernel; the same posterior and scalars
    assert torch.equal(fwd.ability_mu, full.ability_mu) and rel_err(fwd.flat[:7].cpu(), full.flat[:7].cpu()) < 1e-6
    # the other form of the same call (three passes <-> experts' sums in the kernel)
    other_flags = ops.DESC_FLAGS ^ _lib.FLAG_COND_THREE_PASS if (ops.DESC_FLAGS & _lib.FLAG_KERNEL_MATRIX) else ops.DESC_FLAGS
    monkeypatch.setattr(ops, 'DESC_FLAGS', other_flags)
    other = run()
    assert rel_err(other.flat[:7].cpu(), full.flat[:7].cpu()) < 2e-6
    assert rel_err(other.flat[8:].cpu(), full.flat[8:].cpu()) < 2e-5
    assert (other.ability_mu - full.ability_mu).abs().max() < 2e-6 and (other.ability_logvar - full.ability_logvar).abs().max() < 2e-6
    n = 64
    ref = T.fused_elbo_ref(table.cpu().double(), item.cpu().double(), resp[:n].cpu().double(), mask[:n].cpu(),
                           eps[:n].cpu().double(), irt_model=irt, ability_dim=A, conditional_posterior=True, mode='kl')
    for k, t in (('ability_mu', full.ability_mu), ('ability_logvar', full.ability_logvar), ('ability', full.ability)):
        assert (t[:n].cpu() - ref[k].float()).abs().max() < 2e-5 * max(1.0, float(ref[k].abs().max())), k


@pytest.mark.parametrize('codes', [False, True], ids=['fp32-rows', 'cell-codes'])
def test_resident_row_counts_change_nothing(codes, monkeypatch):
    """Rows of more than 1024 items under the unconditional posterior: a matrix the process calls with a second time is counted once
    (ops._resident_row_counts -> vibo_elbo_fwd_bwd_counts) and the count pass in front of the panels goes away -- the whole matrix and
    minibatches gathered from it give the bits of the counting call."""
    irt, A, P, I = 2, 2, 3000, 2500
    d = dev()
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    resp, mask, table, item, eps = _device_problem(irt, A, P, I, 0.2, seed=77, cond=False)
    if codes:
        r2 = m2 = ops.pack_cell_codes(resp, mask).codes
        code = _lib.MASK_CODES
    else:
        rp, mp = ops.pad_rows(resp, mask)
        r2, m2, code = ops.prepare_rows(rp, mp)
    rows = torch.randperm(P, device=d, generator=torch.Generator(device=d).manual_seed(3))[:700].contiguous()

    def run(row_index=None):
        e = eps if row_index is None else eps[row_index].contiguous()
        B = P if row_index is None else int(row_index.numel())
        out = ops._hip_launch_elbo(spec, r2, m2, code, row_index, table, item, e, None, _lib.REG_KL, True, B)
        torch.cuda.synchronize()
        return out

    monkeypatch.setattr(ops, 'ROW_COUNT_CACHE', False)
    full0, mb0 = run(), run(rows)
    monkeypatch.setattr(ops, 'ROW_COUNT_CACHE', True)
    run()                                        # first sighting: still the counting call
    assert r2._vibo_row_counts[3] is None
    full1, mb1 = run(), run(rows)                # second: counted once, handed over from here on
    assert r2._vibo_row_counts[3] is not None
    for a, b in ((full0, full1), (mb0, mb1)):
        assert torch.equal(a.flat, b.flat) and torch.equal(a.ability_mu, b.ability_mu) and torch.equal(a.ability, b.ability)


def test_resident_row_counts_through_the_module(monkeypatch):
    """The same through the module path (models.py:337-354 -> elbo): a resident split's tensors come back every step -- the bool mask's
    uint8 view is one object per mask (ops.prepare_mask), so the second step finds the matrix in ops._resident_row_counts -- and the
    losses / gradients of minibatch steps are those of the counting calls, bit for bit."""
    from vibo_amd.torch_core.models import VIBO_2PL
    A, P, I = 2, 900, 2100
    d = dev()
    resp, mask, _, _, _ = _device_problem(2, A, P, I, 0.2, seed=78, cond=False)
    mask = mask.bool()
    torch.manual_seed(5)
    model = VIBO_2PL(A, I, ability_merge='product').to(d)
    rows = [torch.randperm(P, device=d, generator=torch.Generator(device=d).manual_seed(k))[:256].contiguous() for k in range(3)]

    def steps():
        out = []
        for k in range(3):
            torch.manual_seed(100 + k)
            model.zero_grad()
            loss = model.elbo_step(resp, mask, row_index=rows[k])
            loss.backward()
            out.append((loss.detach().clone(), [p_.grad.detach().clone() for p_ in model.parameters()]))
        torch.cuda.synchronize()
        return out

    monkeypatch.setattr(ops, 'ROW_COUNT_CACHE', False)
    ref = steps()
    monkeypatch.setattr(ops, 'ROW_COUNT_CACHE', True)
    got = steps()
    assert getattr(resp, '_vibo_row_counts', None) is not None and resp._vibo_row_counts[3] is not None, 'the resident matrix was not recognised'
    for (l0, g0), (l1, g1) in zip(ref, got):
        assert torch.equal(l0, l1)
        for a, b in zip(g0, g1):
            assert torch.equal(a, b)


def test_config4_shape_at_full_size():
    """BASELINE configs[3]'s matrix shape at its full size (CritLangAcq: 535 598 persons x 95 items, 2PL, ability_dim 1,
    --artificial-missing-perc 0.2; datasets.py:283-440, masked log-likelihood models.py:596-629): narrow rows with padded
    16-byte strides -- bitwise reproducible, additive over two person shards, invariant under a person permutation (in-kernel
    gather), equal to the CPU oracle on a slice."""
    irt, A, P, I = 2, 1, 535_598, 95
    d = dev()
    spec = ElboSpec(irt_model=irt, ability_dim=A)
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
    perm = torch.randperm(P, device=d, generator=torch.Generator(device=d).manual_seed(9))
    pg = run(row_index=perm)
    assert rel_err(pg.flat[:7].cpu(), full.flat[:7].cpu()) < 1e-5
    assert rel_err(pg.flat[8:].cpu(), full.flat[8:].cpu()) < 1e-4
    assert (pg.ability_mu - full.ability_mu[perm]).abs().max() < 1e-6
    n = 512
    ref = T.fused_elbo_ref(table.cpu().double(), item.cpu().double(), resp[:n].cpu().double(), mask[:n].cpu(),
                           eps[:n].cpu().double(), irt_model=irt, ability_dim=A, mode='kl')
    compare_raw(run(slice(0, n)), ref, (I, A + 1))
    assert (full.ability_mu[:n].cpu() - ref['ability_mu'].float()).abs().max() < 2e-5


@pytest.mark.parametrize('A,I,codes', [(8, 1000, False), (6, 1100, False), (8, 420, True), (5, 2100, True)])
def test_wide_conditional_posterior_is_deterministic(A, I, codes):
    """conditional_posterior with ability_dim 5..8 (utils.py:85-113 product of per-cell experts, models.py:607-640):
    the per-row statistics / per-item gradient passes run once per 4 ability dims, so these widths take the same
    block-partial + fixed-order-finalize path as A <= 4 -- no float atomics: two runs agree bit for bit, and with the oracle."""
    irt, B = 2, 900
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=True)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.25, seed=A * I, cond=True)
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(), irt_model=irt,
                           ability_dim=A, mode='kl', conditional_posterior=True)
    d = dev()
    if codes:
        cells = ops.pack_cell_codes(resp.to(d), mask.to(d)).codes
        run = lambda: ops._hip_launch_elbo(spec, cells, cells, _lib.MASK_CODES, None, table.to(d).contiguous(),
                                           item.to(d).contiguous(), eps.to(d).contiguous(), None, _lib.REG_KL, True, B)
    else:
        run = lambda: run_kernel(spec, resp, mask, table, item, eps)
    a, b = run(), run()
    torch.cuda.synchronize()
    assert torch.equal(a.flat, b.flat) and torch.equal(a.grad_table(0), b.grad_table(0))
    compare_raw(a, ref, (I, A + 1))


MATRIX_PIPE_POSTERIOR = [
    # irt, A, B, I, flows, gather, codes, drop
    (2, 1, 4100, 1000, 0, False, True, False),     # caller's cell codes: both passes on the matrix pipe
    (2, 8, 4100, 1000, 0, True, True, False),      # rows through row_index, 2 N-tiles of coefficients
    (3, 3, 4500, 95, 2, False, False, True),       # fp32 rows, 3 dims: VALU pre pass + matrix-pipe post pass; one whole step + a 31-item tail
    (2, 2, 4100, 200, 0, False, False, False),     # fp32 rows, 2 dims: VALU pre pass (emits the codes) + matrix-pipe post pass
    (2, 5, 4200, 1500, 0, True, False, False),     # fp32 rows, 5 dims: count-and-emit pass in front of the matrix-pipe pre pass; two panels
    (1, 4, 5000, 63, 0, False, True, False),       # fewer than 64 items: only the partial step
    (2, 1, 4096, 64, 0, False, True, False),       # exactly one whole step, nothing else
    (2, 8, 16, 1000, 0, False, True, False),       # the CLI's default minibatch: one partly filled M-tile
    (3, 3, 77, 95, 2, True, False, True),          # fp32 rows, small gathered minibatch
    # fp32 rows at 5+ dims: the forward contraction reads the fp32 cells itself and leaves the codes behind (cm_forward_fp32_kernel)
    (2, 8, 300, 1000, 0, True, False, False),      # 8 dims (observed counts from the code words), gathered rows, 15 steps + a 40-item tail
    (2, 6, 16, 128, 0, False, False, True),        # two whole steps and nothing else, 16 persons
    (3, 5, 100, 40, 2, False, False, False),       # fewer than 64 items: only the partial step
    (2, 8, 4100, 1000, 0, False, False, False),    # in order: must equal the same call on packed cell codes bit for bit (below)
]


@pytest.mark.parametrize('irt,A,B,I,n_flows,gather,codes,drop', MATRIX_PIPE_POSTERIOR)
def test_matrix_pipe_passes_of_the_conditional_posterior(irt, A, B, I, n_flows, gather, codes, drop, monkeypatch):
    """From a call size that depends on ability_dim (any size at 5+ dims) the two extra passes of --conditional-posterior (the experts' precision sums per person,
    models.py:664-710 + utils.py:105-113, and the scatter of the table gradient) run as one-hot x table contractions on the
    matrix pipe (vibo_cmean.hip): against the fp64 oracle, bit-identical between runs, and in agreement with the VALU passes
    (VIBO_FLAG_COND_VALU) the smaller calls keep.  VIBO_FLAG_COND_MATRIX pins the matrix-pipe form for the test."""
    kernel = ops.DESC_FLAGS & (_lib.FLAG_KERNEL_MATRIX | _lib.FLAG_KERNEL_VALU)
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=True, n_flows=n_flows, drop_missing=drop)
    resp, mask, table, item, eps = random_problem(irt, A, B + 9, I, 0.2, seed=A * I + B, cond=True)
    if drop:
        mask[:, 0] = 1
        resp[:, 0] = resp[:, 0].clamp(min=0)
    g = torch.Generator().manual_seed(I)
    flow = torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5 if n_flows else None
    flows = [(f_[:A].double(), f_[A:2 * A].double(), f_[2 * A:].double()) for f_ in flow] if n_flows else None
    rows = torch.randperm(B + 9, generator=g)[:B] if gather else torch.arange(B)
    ref = T.fused_elbo_ref(table.double(), item.double(), resp[rows].double(), mask[rows], eps.double()[:B], irt_model=irt,
                           ability_dim=A, conditional_posterior=True, replace_missing_with_prior=not drop,
                           mode='sampled' if n_flows else 'kl', flow_uhat_w_b=flows)
    d = dev()
    reg = _lib.REG_SAMPLED if n_flows else _lib.REG_KL
    ri = rows.to(d) if gather else None
    if not gather:
        resp, mask = resp[:B], mask[:B]
    if codes:
        r = m = ops.pack_cell_codes(resp.to(d), mask.to(d)).codes
        code = _lib.MASK_CODES
    else:
        r_, m_ = ops.pad_rows(resp.to(d), mask.bool().to(d))          # 16-byte rows: the row-split path (95 items: padded strides)
        r, m, code = ops.prepare_rows(r_, m_)
    args = (spec, r, m, code, ri, table.to(d).contiguous(), item.to(d).contiguous(), eps[:B].to(d).contiguous(),
            flow.to(d).contiguous() if flow is not None else None, reg)

    def run(flags, want_grad=True):
        monkeypatch.setattr(ops, 'DESC_FLAGS', kernel | flags)
        out = ops._hip_launch_elbo(*args, want_grad, B)
        torch.cuda.synchronize()
        return out

    a, b = run(_lib.FLAG_COND_MATRIX), run(_lib.FLAG_COND_MATRIX)
    assert torch.equal(a.flat, b.flat) and torch.equal(a.ability_mu, b.ability_mu)
    compare_raw(a, ref, (I, spec.item_dim))
    v = run(_lib.FLAG_COND_VALU)
    assert (a.ability_mu - v.ability_mu).abs().max() < 1e-5 * max(1.0, float(v.ability_mu.abs().max()))
    assert (a.ability_logvar - v.ability_logvar).abs().max() < 1e-5 * max(1.0, float(v.ability_logvar.abs().max()))
    for s_ in range(2):
        assert rel_err(a.grad_table(s_), v.grad_table(s_)) < 2e-5
    # vibo_encode (inference: the posterior alone) takes the same pre pass
    monkeypatch.setattr(ops, 'DESC_FLAGS', kernel | _lib.FLAG_COND_MATRIX)
    emu, elv = ops._hip_encode(spec, r, m, code, ri, table.to(d).contiguous(), B)
    assert (emu - a.ability_mu).abs().max() < 2e-6 * max(1.0, float(a.ability_mu.abs().max()))
    assert (elv - a.ability_logvar).abs().max() < 2e-6 * max(1.0, float(a.ability_logvar.abs().max()))
    f = run(_lib.FLAG_COND_MATRIX, want_grad=False)  # forward only: the same posterior and scalars
    assert torch.equal(f.ability_mu, a.ability_mu) and torch.equal(f.ability_logvar, a.ability_logvar)
    assert rel_err(f.scalars[_lib.S_LL], a.scalars[_lib.S_LL]) < 1e-6
    if not codes and A >= 5 and not gather:
        # Format R at 5+ dims = the same three passes on the codes the first one emits: bit-identical to the Format P call
        pc = ops.pack_cell_codes(resp.to(d), mask.to(d)).codes
        monkeypatch.setattr(ops, 'DESC_FLAGS', kernel | _lib.FLAG_COND_MATRIX)
        c = ops._hip_launch_elbo(spec, pc, pc, _lib.MASK_CODES, None, *args[5:], True, B)
        torch.cuda.synchronize()
        assert torch.equal(c.ability_mu, a.ability_mu) and torch.equal(c.ability_logvar, a.ability_logvar) and torch.equal(c.flat, a.flat)


@pytest.mark.parametrize('irt,A,I,cond,n_flows', [(2, 1, 95, False, 0), (2, 8, 1003, False, 0), (3, 2, 333, False, 2),
                                                   (2, 1, 2501, False, 0), (2, 2, 201, True, 0)])
def test_ragged_item_count_with_padded_row_strides(irt, A, I, cond, n_flows):
    """I % 4 != 0 (CritLangAcq: 95 items): rows padded to 16 bytes by ops.pad_rows take the row-split kernels; the
    padding cells (and, for a column-sliced view, the neighbouring columns) are read but never interpreted."""
    B = 77
    spec = ElboSpec(irt_model=irt, ability_dim=A, conditional=cond, n_flows=n_flows)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.2, seed=I + A, cond=cond)
    g = torch.Generator().manual_seed(I)
    flow = torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5 if n_flows else None
    flows = [(f[:A].double(), f[A:2 * A].double(), f[2 * A:].double()) for f in flow] if n_flows else None
    mode = 'sampled' if n_flows else 'kl'
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(), irt_model=irt,
                           ability_dim=A, conditional_posterior=cond, replace_missing_with_prior=True, mode=mode,
                           flow_uhat_w_b=flows)
    d = dev()
    # (a) padded copies; (b) a column slice of a wider matrix whose extra columns hold garbage that must be ignored
    wide_r = torch.full((B, I + 9), 1.0)
    wide_m = torch.ones(B, I + 9, dtype=torch.bool)
    wide_r[:, :I] = resp
    wide_m[:, :I] = mask.bool()
    cases = [ops.pad_rows(resp.to(d), mask.bool().to(d))]
    if (I + 9) % 4 == 0:
        cases.append((wide_r.to(d)[:, :I], wide_m.to(d)[:, :I]))
    for r_, m_ in cases:
        assert r_.stride(0) % 4 == 0 and r_.stride(0) >= I
        r = ops.prepare_response(r_)
        m, code = ops.prepare_mask(m_)
        assert r.stride(0) == r_.stride(0) and m.stride(0) == m_.stride(0)          # read in place, not copied
        raw = ops._hip_launch_elbo(spec, r, m, code, None, table.to(d).contiguous(), item.to(d).contiguous(),
                                   eps.to(d).contiguous(), flow.to(d).contiguous() if flow is not None else None,
                                   _lib.REG_SAMPLED if n_flows else _lib.REG_KL, True, B)
        torch.cuda.synchronize()
        compare_raw(raw, ref, (I, spec.item_dim))


# ---------------------------------------------------------------------------
# (4) multi-sample forward (log_marginal's loop body, models.py:445-504)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('irt,A,B,I,S,n_flows', [(2, 1, 100, 1000, 7, 0), (2, 8, 130, 1000, 5, 0), (3, 2, 77, 600, 4, 2),
                                                 (1, 4, 33, 332, 3, 0), (2, 2, 50, 2500, 6, 4), (2, 3, 41, 95, 2, 0)])
def test_multi_sample_forward_equals_single_launches(irt, A, B, I, S, n_flows):
    """vibo_elbo_multi_forward (2 / 4 samples per pass over the rows) returns the heads of S separate forward launches."""
    spec = ElboSpec(irt_model=irt, ability_dim=A, n_flows=n_flows)
    resp, mask, table, _, _ = random_problem(irt, A, B, I, 0.15, seed=S * 100 + I)
    g = torch.Generator().manual_seed(I + S)
    items = torch.randn(S, I, spec.item_dim, generator=g)
    eps = torch.randn(S, B, A, generator=g)
    flow = torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5 if n_flows else None
    d = dev()
    r_, m_ = ops.pad_rows(resp.to(d), mask.bool().to(d))
    r = ops.prepare_response(r_)
    m, code = ops.prepare_mask(m_)
    fl = flow.to(d).contiguous() if flow is not None else None
    sc = ops._hip_multi_forward(spec, r, m, code, None, table.to(d).contiguous(), items.to(d).contiguous(),
                                eps.to(d).contiguous(), fl, _lib.REG_SAMPLED, B)
    assert sc is not None and sc.shape == (S, _lib.NUM_SCALARS)
    for s in range(S):
        one = ops._hip_launch_elbo(spec, r, m, code, None, table.to(d).contiguous(), items[s].to(d).contiguous(),
                                   eps[s].to(d).contiguous(), fl, _lib.REG_SAMPLED, False, B)
        a, b = sc[s, :7].cpu().double(), one.scalars[:7].cpu().double()
        assert float((a - b).abs().max()) < 2e-6 * max(1.0, float(b.abs().max())), (s, a, b)


@pytest.mark.parametrize('name', ['logmarg_2pl_a2', 'logmarg_3pl_a1_cond_flows2'])
def test_log_marginal_golden_through_module(name):
    import os
    from conftest import GOLDEN_DIR, Golden
    g = Golden(os.path.join(GOLDEN_DIR, name + '.npz'))
    d = dev()
    model = build_model(g).to(d)
    logp = model.log_marginal(g.response.to(d).unsqueeze(2), g.mask.to(d).bool().unsqueeze(2),
                              num_samples=g.meta['num_samples'], eps_item=g.eps_item.to(d), eps_ability=g.eps_ability.to(d))
    ref = float(g.out['logp'])
    assert abs(float(logp) - ref) < 1e-4 * max(1.0, abs(ref))


@pytest.mark.parametrize('irt,A,B,I,S', [(2, 1, 100, 95, 7), (3, 2, 33, 200, 3), (1, 4, 9, 1000, 5), (2, 8, 17, 64, 2)])
def test_decode_mean_kernel(irt, A, B, I, S):
    """vibo_decode_mean = mean over S draws of decode (posterior predictive, vibo.py:363-390 / 504-548)."""
    spec = ElboSpec(irt_model=irt, ability_dim=A)
    g = torch.Generator().manual_seed(B + I)
    ab = torch.randn(S, B, A, generator=g)
    it = torch.randn(S, I, spec.item_dim, generator=g)
    ref = torch.stack([O.irt_link(irt, ab[s].double(), it[s].double()) for s in range(S)]).mean(0)
    out = ops.decode_probs_mean(spec, ab.to(dev()), it.to(dev())).cpu().double()
    assert float((out - ref).abs().max()) < 2e-6


def test_encode_and_decode_on_a_flow_model():
    """model.encode / decode / posterior-predictive mean do not involve the flows, but must accept a flow model's spec
    (found by tools/fuzz_parity.py: the descriptor used to be rejected)."""
    irt, A, B, I = 2, 2, 50, 200
    spec = ElboSpec(irt_model=irt, ability_dim=A, n_flows=4)
    resp, mask, table, item, eps = random_problem(irt, A, B, I, 0.1, seed=4)
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(), irt_model=irt, ability_dim=A,
                           mode='kl', want_grad=False)
    d = dev()
    mu, lv = ops.encode_posterior(spec, table.to(d), resp.to(d), mask.bool().to(d))
    assert (mu.cpu() - ref['ability_mu'].float()).abs().max() < 2e-5
    pr = ops.decode_probs(spec, ref['ability'].float().to(d), item.to(d)).cpu()
    assert (pr - O.irt_link(irt, ref['ability'].float(), item)).abs().max() < 2e-6
    prm = ops.decode_probs_mean(spec, ref['ability'].float().to(d)[None], item.to(d)[None]).cpu()
    assert (prm - pr).abs().max() < 2e-6


@pytest.mark.parametrize('I,A', [(50, 2), (95, 1), (1030, 3)])
def test_conditional_posterior_does_not_touch_memory_behind_the_table(I, A):
    """Regression (found by the trained-model parity run: NaN in epoch 12 of a conditional-posterior training with 50 items):
    the last 4-item chunk of a row whose item count is not a multiple of 4 reaches past the per-item expert table; whatever
    lies there (here: NaN) must not enter the sums, not even multiplied by a zero weight."""
    d = dev()
    B = 16
    spec = ElboSpec(irt_model=2, ability_dim=A, conditional=True)
    resp, mask, table, item, eps = random_problem(2, A, B, I, 0.2, seed=I + A, cond=True)
    ref = T.fused_elbo_ref(table.double(), item.double(), resp.double(), mask, eps.double(), irt_model=2, ability_dim=A,
                           conditional_posterior=True, mode='kl')
    n = table.numel()
    buf = torch.full((n + 256,), float('nan'), device=d)
    buf[:n] = table.to(d).reshape(-1)
    tab = buf[:n].view(2, I, 2 * A)
    ibuf = torch.full((item.numel() + 256,), float('nan'), device=d)
    ibuf[:item.numel()] = item.to(d).reshape(-1)
    it = ibuf[:item.numel()].view(I, spec.item_dim)
    r_, m_ = ops.pad_rows(resp.to(d), mask.bool().to(d))
    for rows in ((r_, m_), (ops.pack_cell_codes(r_, m_), None)):
        r, m, code = ops.prepare_rows(*rows)
        raw = ops._hip_launch_elbo(spec, r, m, code, None, tab, it, eps.to(d), None, _lib.REG_KL, True, B)
        assert torch.isfinite(raw.flat).all() and torch.isfinite(raw.ability_mu).all()
        compare_raw(raw, ref, (I, spec.item_dim))
        mu, lv = ops._hip_encode(spec, r, m, code, None, tab, B)
        assert torch.isfinite(mu).all() and (mu.cpu() - ref['ability_mu'].float()).abs().max() < 2e-5
