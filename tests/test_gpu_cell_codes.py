"""Format P: one byte per cell (VIBO_MASK_CODES) must give the results of the reference layout
(fp32 responses + mask bytes), bit for bit (the kernels see the same fp8 codes either way and reduce in a fixed
order), through every entry point that reads rows."""
import copy

import pytest
import torch

from oracle import vibo_oracle as O
from vibo_amd import _lib, ops
from vibo_amd.ops import ElboSpec
from vibo_amd.torch_core.models import VIBO_2PL, VIBO_3PL
from vibo_amd.trainer import FusedTrainer

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
dev = torch.device('cuda:0')


def problem(irt, A, B, I, cond, n_flows, missing=0.2, seed=0):
    g = torch.Generator().manual_seed(seed + 7 * I + A)
    resp, mask = O.simulate_responses(irt, B, I, A, generator=g, missing_frac=missing)
    spec = ElboSpec(irt_model=irt, ability_dim=A, n_flows=n_flows, conditional=cond)
    table = (torch.randn(*spec.table_shape(I), generator=g) * 0.6).to(dev)
    item = (torch.randn(I, spec.item_dim, generator=g) * 0.7).to(dev)
    eps = torch.randn(B, A, generator=g).to(dev)
    flow = (torch.randn(n_flows, 2 * A + 1, generator=g) * 0.5).to(dev) if n_flows else None
    return spec, resp.to(dev), mask.bool().to(dev), table, item, eps, flow


@pytest.mark.parametrize('mask_kind', ['bool', 'int64', 'none'])
@pytest.mark.parametrize('I', [1, 95, 100, 1028])
def test_pack_codes_matches_the_layout_contract(I, mask_kind):
    g = torch.Generator().manual_seed(I)
    resp, mask = O.simulate_responses(2, 37, I, 1, generator=g, missing_frac=0.3)
    resp, mask = resp.to(dev), mask.bool().to(dev)
    if mask_kind == 'none':
        resp = resp.clamp(min=0)            # no mask: every cell is an answer
    m = {'bool': mask, 'int64': mask.long(), 'none': None}[mask_kind]
    cc = ops.pack_cell_codes(resp.unsqueeze(2), m.unsqueeze(2) if m is not None else None)   # [P,I,1] as the loaders give it
    want = torch.where(mask if m is not None else torch.ones_like(mask), (resp == 1).to(torch.uint8),
                       torch.full_like(resp, 2, dtype=torch.uint8))
    assert cc.codes.shape == (37, I) and cc.codes.stride(0) % 4 == 0
    assert torch.equal(cc.codes, want)
    stride = cc.codes.stride(0)
    pad = cc.codes.as_strided((37, stride), (stride, 1))[:, I:]
    assert bool((pad == 2).all())                       # padding cells read as missing
    r2, m2 = cc.unpack()
    assert torch.equal(m2, mask if m is not None else torch.ones_like(mask))
    assert torch.equal(r2[m2], resp[m2])
    sub = cc.rows(torch.tensor([5, 0, 36], device=dev))
    assert torch.equal(sub.codes, want[[5, 0, 36]]) and sub.codes.stride(0) % 4 == 0


CASES = [
    # irt, A, B, I, cond, flows, drop, gather
    (2, 1, 200, 1000, False, 0, False, False),
    (2, 8, 130, 1000, False, 0, False, True),
    (2, 3, 77, 600, False, 0, True, False),
    (3, 2, 64, 304, False, 0, False, False),
    (1, 4, 50, 144, False, 0, False, True),
    (2, 1, 41, 95, False, 0, False, False),       # ragged: last chunk partly padding
    (2, 5, 33, 7, False, 0, False, False),
    (2, 2, 90, 1000, False, 4, False, False),     # planar flows
    (3, 8, 64, 1028, False, 2, False, True),      # panels + flows
    (2, 1, 64, 2500, False, 0, False, False),     # panels: row-count pass reads the codes
    (2, 8, 17, 10000, False, 0, True, True),
    (2, 1, 100, 100, True, 0, False, False),      # conditional posterior: cond_pre / split / cond_post
    (3, 4, 64, 1500, True, 0, False, True),
    (2, 2, 30, 95, True, 2, True, False),
]


@pytest.mark.parametrize('irt,A,B,I,cond,n_flows,drop,gather', CASES)
@pytest.mark.parametrize('want_grad', [True, False])
def test_codes_equal_reference_layout(irt, A, B, I, cond, n_flows, drop, gather, want_grad):
    spec, resp, mask, table, item, eps, flow = problem(irt, A, B + 6, I, cond, n_flows)
    spec = ElboSpec(irt_model=irt, ability_dim=A, n_flows=n_flows, conditional=cond, drop_missing=drop)
    if drop:
        mask[:, 0] = True
        resp[:, 0] = resp[:, 0].clamp(min=0)
    rows = torch.randperm(B + 6, generator=torch.Generator().manual_seed(B + I))[:B].to(dev) if gather else None
    if rows is None:
        resp, mask = resp[:B], mask[:B]
    reg = _lib.REG_SAMPLED if n_flows else _lib.REG_KL
    r_, m_ = ops.pad_rows(resp, mask)
    r, m, code = ops.prepare_rows(r_, m_)
    ref = ops._hip_launch_elbo(spec, r, m, code, rows, table, item, eps, flow, reg, want_grad, B)
    c, cm, ccode = ops.prepare_rows(ops.pack_cell_codes(resp, mask), None)
    assert ccode == _lib.MASK_CODES
    got = ops._hip_launch_elbo(spec, c, cm, ccode, rows, table, item, eps, flow, reg, want_grad, B)
    n = got.flat.numel() if want_grad else _lib.NUM_SCALARS
    pairs = [(got.flat[:n], ref.flat[:n]), (got.ability_mu, ref.ability_mu), (got.ability_logvar, ref.ability_logvar),
             (got.ability, ref.ability)]
    if n_flows:
        pairs += [(got.ability_k, ref.ability_k), (got.ability_ladj, ref.ability_ladj)]
    # same fp8 code words, same fixed-order reductions: bit for bit -- except where the two calls take different forms of the
    # conditional posterior's pre pass (cell codes: matrix pipe when pinned or planned; fp32 rows up to 4 dims: the VALU pass)
    lib, ct = _lib.load(), __import__('ctypes')
    forms = [0, 0]
    if cond:
        forms = [lib.vibo_plan_cond_passes(ct.byref(ops._make_desc(spec, B, I, code, reg, want_grad, r.stride(0), m.stride(0) if m is not None else 0))),
                 lib.vibo_plan_cond_passes(ct.byref(ops._make_desc(spec, B, I, ccode, reg, want_grad, c.stride(0), cm.stride(0))))]
        assert min(forms) >= 0
    for k, (x, y) in enumerate(pairs):
        if forms[0] == forms[1]:
            if k == 0:       # (the two diagnostic sums log q(theta_0), log p(theta_K) of the KL-mode call: the gathered fp32 and the
                             #  cell-code instantiation of the matrix kernel contract their per-person terms differently -- 1 ulp of
                             #  the sum on ~6 % of row subsets; everything the loss uses is bit-identical)
                keep = torch.ones_like(x, dtype=torch.bool)
                keep[_lib.S_LOGQ0] = keep[_lib.S_LOGP] = False
                assert torch.equal(x[keep], y[keep]), k
                assert torch.allclose(x[~keep], y[~keep], rtol=1e-6, atol=0), k
                continue
            assert torch.equal(x, y), k
        else:
            assert (x - y).abs().max() <= 3e-6 * max(1.0, float(y.abs().max())), k
    # forward-only posterior (encode) and the multi-sample forward read the same rows
    emu, elv = ops._hip_encode(spec, c, cm, ccode, rows, table, B)
    rmu, rlv = ops._hip_encode(spec, r, m, code, rows, table, B)
    # (the two calls may take different forms of the pre pass: a few ulps of log sigma^2 ~ -7)
    assert torch.allclose(emu, rmu, rtol=0, atol=1e-6) and torch.allclose(elv, rlv, rtol=1e-6, atol=1e-6)
    if not cond and not want_grad:
        items = torch.stack([item, item * 0.5, item + 0.1]).contiguous()
        epss = torch.stack([eps, -eps, eps * 0.3]).contiguous()
        a = ops._hip_multi_forward(spec, c, cm, ccode, rows, table, items, epss, flow, _lib.REG_SAMPLED, B)
        b = ops._hip_multi_forward(spec, r, m, code, rows, table, items, epss, flow, _lib.REG_SAMPLED, B)
        assert a is not None and torch.equal(a, b)


def test_codes_outside_the_row_split_paths_are_refused():
    # (fewer than 4 items; the conditional posterior with ability_dim > 4 used to be refused too and now runs on the
    #  row-split path: test_gpu_parity.test_wide_conditional_posterior_is_deterministic)
    for (A, I, cond) in [(1, 3, False)]:
        spec, resp, mask, table, item, eps, _ = problem(2, A, 16, I, cond, 0)
        c, cm, ccode = ops.prepare_rows(ops.pack_cell_codes(resp, mask), None)
        with pytest.raises(RuntimeError, match='cell codes'):
            ops._hip_launch_elbo(spec, c, cm, ccode, None, table, item, eps, None, _lib.REG_KL, True, 16)
        with pytest.raises(RuntimeError, match='cell codes'):
            ops._hip_encode(spec, c, cm, ccode, None, table, 16)
    with pytest.raises(ValueError):
        ops.prepare_rows(ops.pack_cell_codes(resp, mask), mask)          # codes carry their own missingness


@pytest.mark.parametrize('cls,A,I,kw', [(VIBO_2PL, 2, 100, {}), (VIBO_3PL, 1, 95, {'n_norm_flows': 2}),
                                        (VIBO_2PL, 3, 260, {'conditional_posterior': True})])
def test_module_and_trainer_on_cell_codes(cls, A, I, kw):
    """Drop-in module (forward -> elbo -> backward, encode, log_marginal) and the fused trainer fed CellCodes rows
    follow the reference-layout run under the same noise."""
    g = torch.Generator().manual_seed(5)
    resp, mask = O.simulate_responses(cls.IRT, 300, I, A, generator=g, missing_frac=0.15)
    resp, mask = resp.to(dev), mask.bool().to(dev)
    codes = ops.pack_cell_codes(resp, mask)
    resp, mask = ops.pad_rows(resp, mask)        # 95 items: padded rows keep the fp32 layout on the row-split path too
    torch.manual_seed(1)
    m_ref = cls(A, I, ability_merge='product', **kw).to(dev)
    m_cod = copy.deepcopy(m_ref)
    rows = torch.randperm(300, generator=g)[:64].to(dev)
    # bit for bit -- except with the conditional posterior, whose extra passes take the form that is faster for the row format
    # (fp32 rows: the VALU pre pass that also emits the codes; cell codes: the matrix pipe from a size that depends on ability_dim)
    tol = 3e-6 if kw.get('conditional_posterior') else 0.0
    for model, (r, m) in ((m_ref, (resp, mask)), (m_cod, (codes, None))):
        torch.manual_seed(11)
        loss = model.elbo_step(r, m, annealing_factor=0.7, row_index=rows)
        loss.backward()
        model._loss = loss.detach()
        torch.manual_seed(12)
        model._enc = model.encode(r, m, row_index=rows)
        torch.manual_seed(13)
        model._lm = model.log_marginal(*((resp[rows], mask[rows]) if m is not None else (codes.rows(rows), None)), num_samples=5)
    assert abs(float(m_cod._loss) - float(m_ref._loss)) <= tol * abs(float(m_ref._loss))
    for (k, a), (_, b) in zip(m_cod.named_parameters(), m_ref.named_parameters()):
        assert (a.grad - b.grad).abs().max() <= tol * max(1.0, float(b.grad.abs().max())) , k
    for a, b in zip(m_cod._enc, m_ref._enc):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
    assert torch.allclose(m_cod._lm, m_ref._lm, rtol=1e-6, atol=1e-4)
    if not kw:
        t_ref, t_cod = FusedTrainer(m_ref, lr=5e-3), FusedTrainer(m_cod, lr=5e-3)
        for step in range(3):
            torch.manual_seed(20 + step)
            l_ref = t_ref.step(resp, mask, beta=1.0, row_index=rows)
            torch.manual_seed(20 + step)
            l_cod = t_cod.step(codes, None, beta=1.0, row_index=rows)
            assert torch.equal(l_ref, l_cod)
        for (k, a), (_, b) in zip(m_cod.state_dict().items(), m_ref.state_dict().items()):
            assert torch.equal(a, b), k
