"""Host side of the fused ELBO op: launches libvibo_hip.so through the C ABI and
wires its outputs into torch.autograd.

Reference seam: VIBO_*PL.forward/encode/decode/elbo of
src/torch_core/models.py:337-443 as called from src/torch_core/vibo.py:237-268.

What stays in PyTorch (tiny, O(I) work): the 2-row (or 2xI-row) encoder MLP that
produces the expert table, the item-side reparameterisation / flows / KL, the
optimizer.  Everything O(B x I) is inside the HIP kernel.
"""
import contextlib
import ctypes
import weakref
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


def item_feat_dim(irt_model, ability_dim):
    """models.py:331-332, 523-524, 538-539."""
    return {1: 1, 2: ability_dim + 1, 3: ability_dim + 2}[int(irt_model)]


@dataclass(frozen=True)
class ElboSpec:
    """Static configuration of one fused-ELBO problem family."""
    irt_model: int
    ability_dim: int
    conditional: bool = False
    drop_missing: bool = False       # --drop-missing (vibo.py:52,217)
    n_flows: int = 0
    given: bool = False              # the caller supplies q(theta_p) per person (--ability-merge mean): VIBO_POSTERIOR_GIVEN

    @property
    def item_dim(self):
        return item_feat_dim(self.irt_model, self.ability_dim)

    def table_shape(self, num_item, num_person=None):
        A = self.ability_dim
        if self.given:
            return (num_person, 2 * A)
        return (2, num_item, 2 * A) if self.conditional else (2, 2 * A)

    def check_supported(self, num_item):
        if not (1 <= self.ability_dim <= _lib.MAX_ABILITY_DIM):
            raise NotImplementedError(
                f'ability_dim={self.ability_dim}: the HIP kernel supports 1..{_lib.MAX_ABILITY_DIM}')
        if not (0 <= self.n_flows <= _lib.MAX_FLOWS):
            raise NotImplementedError(f'n_norm_flows={self.n_flows}: supported 0..{_lib.MAX_FLOWS}')


@dataclass
class RawElbo:
    """Everything one kernel call produces (device tensors)."""
    flat: torch.Tensor              # [8 scalars | grad_table(2*T) | grad_item(I*D) | grad_flow(2*F)]
    n_table: int
    n_item: int
    n_flow: int
    table_shape: tuple
    ability_mu: torch.Tensor
    ability_logvar: torch.Tensor
    ability: torch.Tensor
    ability_k: Optional[torch.Tensor]
    ability_ladj: Optional[torch.Tensor]
    workspace: Optional[torch.Tensor] = None      # folded train step: the partial records vibo_train_epilogue_fused sums

    @property
    def scalars(self):
        return self.flat[:_lib.NUM_SCALARS]

    def grad_table(self, s):
        o = _lib.NUM_SCALARS + s * self.n_table
        return self.flat[o:o + self.n_table].view(self.table_shape)

    def grad_item(self, shape):
        o = _lib.NUM_SCALARS + 2 * self.n_table
        return self.flat[o:o + self.n_item].view(shape)

    def grad_flow(self, s):
        o = _lib.NUM_SCALARS + 2 * self.n_table + self.n_item + s * self.n_flow
        return self.flat[o:o + self.n_flow]


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_U8_VIEW_CACHE = []        # [(weakref to a bool mask, its uint8 view)], newest last, at most 4; an entry dies with its mask
_I64_MASK_CACHE = []       # [(weakref to the int64 tensor, its _version, uint8 copy)], newest last, at most 4


def prepare_mask(mask, keep_int64=False):
    """-> (2-D mask tensor the kernel can read, VIBO_MASK_* code).

    bool / uint8 masks (datasets.py:938 yields bool) are read in place, strided rows included.  int64 masks (the
    reference loop converts with `.long()`, vibo.py:240) cost 8 B per cell and are only understood by the fallback
    kernels, so they are narrowed to uint8 once per tensor (identity-cached: a resident mask is converted a single
    time); `keep_int64=True` hands them to the library unchanged (VIBO_MASK_I64)."""
    if mask is None:
        return None, _lib.MASK_NONE
    if mask.dim() == 3:
        mask = mask.squeeze(2)
    if mask.dtype == torch.bool or mask.dtype == torch.uint8:
        # rows may be strided (e.g. padded to 16 bytes, see pad_rows); only the cells must be unit-stride
        if mask.stride(-1) != 1:
            mask = mask.contiguous()
        if mask.dtype == torch.uint8:
            return mask, _lib.MASK_U8
        # the uint8 view of a bool mask is ONE object per caller tensor while that tensor lives (a view is a new tensor object per
        # call otherwise, and what is kept per resident matrix -- _resident_row_counts -- goes by the objects' identity)
        for ref, view in _U8_VIEW_CACHE:
            if ref() is mask:
                return view, _lib.MASK_U8
        view = mask.view(torch.uint8)
        key = weakref.ref(mask, lambda r: _U8_VIEW_CACHE.__setitem__(slice(None), [e for e in _U8_VIEW_CACHE if e[0] is not r]))
        _U8_VIEW_CACHE[:] = _U8_VIEW_CACHE[-3:]
        _U8_VIEW_CACHE.append((key, view))
        return view, _lib.MASK_U8
    if mask.dtype == torch.int64 and keep_int64:
        return mask.contiguous(), _lib.MASK_I64
    if mask.dtype == torch.int64:
        for ref, version, m8 in _I64_MASK_CACHE:
            if ref() is mask and version == mask._version:
                return m8, _lib.MASK_U8
        m8 = (mask != 0).contiguous().view(torch.uint8)
        _I64_MASK_CACHE[:] = [e for e in _I64_MASK_CACHE if e[0]() is not None and e[0]() is not mask][-3:]
        _I64_MASK_CACHE.append((weakref.ref(mask), mask._version, m8))
        return m8, _lib.MASK_U8
    return (mask != 0).contiguous().view(torch.uint8), _lib.MASK_U8


class CellCodes:
    """"Format P" rows: one byte per cell holding the whole cell (0 = answered wrong, 1 = answered right, 2 = missing;
    VIBO_MASK_CODES of include/vibo_hip.h) instead of an fp32 response plus a mask byte -- 1 B instead of 5 B of HBM
    per cell.  `codes` is a [P, I] uint8 view of rows whose stride is a multiple of 4 cells (pack_cell_codes pads to whole 16-
    or 64-byte pieces, which the matrix-pipe passes of the conditional posterior read fastest).  Passed wherever a `response`
    tensor goes (with mask=None): model.forward / elbo_step / encode / log_marginal, FusedTrainer.step, fused_elbo."""
    __slots__ = ('codes',)

    def __init__(self, codes):
        if codes.dtype != torch.uint8 or codes.dim() != 2 or codes.stride(1) != 1 or codes.stride(0) % 4 != 0:
            raise ValueError('CellCodes: [P, I] uint8 rows with a stride that is a multiple of 4 (see pack_cell_codes)')
        self.codes = codes

    shape = property(lambda self: self.codes.shape)
    device = property(lambda self: self.codes.device)

    def dim(self):
        return 2

    def __len__(self):
        return self.codes.shape[0]

    def rows(self, index):
        """CellCodes of the persons `index` (a copy, rows still padded to a multiple of 4 cells)."""
        c = self.codes
        stride = c.stride(0)
        padded = c.as_strided((c.shape[0], stride), (stride, 1)) if c.shape[1] != stride else c
        return CellCodes(padded[index][:, :c.shape[1]])

    def unpack(self):
        """-> (response fp32 [P, I] with -1 at missing cells, mask bool [P, I]) as the reference stores them
        (datasets.py:928-940)."""
        c = self.codes
        mask = c != 2
        return torch.where(mask, c.float(), c.new_full((), -1, dtype=torch.float32)), mask


def pack_cell_codes(response, mask):
    """Repack device-resident [P, I(,1)] responses (+ mask or None) into CellCodes with vibo_pack_codes (one pass)."""
    lib = _lib.load()
    response = prepare_response(response)
    mask, code = prepare_mask(mask, keep_int64=True)
    _require_device(response, mask)
    P, I = response.shape
    # row stride: whole 64-byte pieces for wide matrices (the matrix-pipe passes of the conditional posterior fetch 64 bytes
    # per row and load: 16-byte aligned, never straddling a 128-byte line), 16-byte rows otherwise
    I4 = (I + 63) // 64 * 64 if I >= 256 else (I + 15) // 16 * 16
    codes = torch.full((P, I4), 2, dtype=torch.uint8, device=response.device)       # padding cells read as missing
    spec = ElboSpec(irt_model=1, ability_dim=1)
    d = _make_desc(spec, P, I, code, _lib.REG_KL, False, response.stride(0), mask.stride(0) if mask is not None else 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream(response.device).cuda_stream)
    rc = lib.vibo_pack_codes(ctypes.byref(d), _ptr(response), _ptr(mask), _ptr(codes), ctypes.c_int64(I4), stream)
    _lib.check(rc, 'vibo_pack_codes')
    return CellCodes(codes[:, :I])


def prepare_rows(response, mask, keep_int64=False):
    """-> (response tensor, mask tensor or None, VIBO_MASK_* code) as the library reads them.  CellCodes rows come back
    as (codes, codes, MASK_CODES): the library ignores the response pointer then."""
    if isinstance(response, CellCodes):
        if mask is not None:
            raise ValueError('CellCodes rows carry their own missingness: pass mask=None')
        return response.codes, response.codes, _lib.MASK_CODES
    response = prepare_response(response)
    mask, code = prepare_mask(mask, keep_int64=keep_int64)
    return response, mask, code


def pad_rows(response, mask):
    """Device-resident copies of a [P, I] response / mask pair whose row strides are padded to a multiple of 4
    cells, returned as [P, I] views.  With I % 4 != 0 (CritLangAcq: 95 items) this keeps every row 16-byte aligned
    so the row-split kernel's vector loads apply; the padding cells are never interpreted (masked in the kernel)."""
    P, I = response.shape
    I4 = (I + 3) // 4 * 4
    if I4 == I:
        return response.contiguous(), (mask.contiguous() if mask is not None else None)
    r = torch.zeros(P, I4, dtype=response.dtype, device=response.device)
    r[:, :I] = response
    m = None
    if mask is not None:
        m = torch.zeros(P, I4, dtype=mask.dtype, device=mask.device)
        m[:, :I] = mask
        m = m[:, :I]
    return r[:, :I], m


def prepare_response(response):
    if response.dim() == 3:
        response = response.squeeze(2)
    if response.dtype != torch.float32:
        response = response.float()
    if response.stride(-1) != 1:
        response = response.contiguous()
    return response


# vibo_desc.flags of every descriptor this module builds (_lib.FLAG_*): 0 = the planner chooses.  Tests and A/B tools
# pin one row-split kernel here; the library itself reads no environment variable.
DESC_FLAGS = 0


@contextlib.contextmanager
def desc_flags(flags):
    """`with ops.desc_flags(_lib.FLAG_KERNEL_MATRIX): ...` -- pin vibo_desc.flags for the calls inside and restore the
    previous value on the way out, also when the body raises (a bare assignment would leave a kernel pinned for the
    rest of the process)."""
    global DESC_FLAGS
    saved, DESC_FLAGS = DESC_FLAGS, int(flags)
    try:
        yield
    finally:
        DESC_FLAGS = saved


class InsituTimer:
    """The fused kernel's duration measured by the kernel itself (vibo_set_insitu_timer, include/vibo_hip.h): usable where HIP
    events are not -- inside a replayed hipGraph -- and without a tracer.  While armed (`with timer:`), every matrix row-split
    launch this host thread enqueues or captures stamps the block; `read()` returns the sums since the last `reset()`.
    A measurement hook for bench.py and tools/: the data path never looks at it."""

    TICK_MS = 1e-5      # s_memrealtime: 100 MHz

    def __init__(self, device):
        self.block = torch.zeros(8, dtype=torch.int64, device=device)
        self.reset()

    def reset(self):
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.block.device).cuda_stream)
        _lib.check(_lib.load().vibo_insitu_timer_reset(_ptr(self.block), stream), 'vibo_insitu_timer_reset')

    def arm(self):
        _lib.check(_lib.load().vibo_set_insitu_timer(_ptr(self.block)), 'vibo_set_insitu_timer')

    @staticmethod
    def disarm():
        _lib.check(_lib.load().vibo_set_insitu_timer(ctypes.c_void_p(0)), 'vibo_set_insitu_timer')

    def __enter__(self):
        self.arm()
        return self

    def __exit__(self, *exc):
        self.disarm()
        return False

    def counters(self):
        """(sum of ticks, launches) since the last reset, after synchronising the device."""
        torch.cuda.synchronize(self.block.device)
        w = self.block.cpu().tolist()
        return w[3], w[4]

    def read(self):
        torch.cuda.synchronize(self.block.device)
        w = self.block.cpu().tolist()
        n = w[4]
        if n <= 0:
            return {'launches': 0}
        k = self.TICK_MS
        return {'launches': n, 'mean_ms': w[3] / n * k, 'min_ms': w[5] * k, 'max_ms': w[6] * k, 'last_ms': w[7] * k,
                'in_flight': w[2] != 0}


def plan_kernel(spec, num_person, num_item, mask_code=_lib.MASK_U8, want_grad=True):
    """Name of the fused kernel the planner picks for a call of this shape (vibo_plan_kernel)."""
    d = _make_desc(spec, num_person, num_item, mask_code, _lib.REG_SAMPLED if spec.n_flows else _lib.REG_KL, want_grad,
                   (num_item + 3) & ~3, (num_item + 3) & ~3)
    k = _lib.load().vibo_plan_kernel(ctypes.byref(d))
    if k < 0:
        _lib.check(k, 'vibo_plan_kernel')
    return _lib.KERNEL_NAMES[k]


def _make_desc(spec, B, I, mask_code, reg_mode, want_grad, resp_stride, mask_stride):
    d = _lib.ViboDesc()
    d.flags = DESC_FLAGS
    d.abi_version = _lib.ABI_VERSION
    d.num_person = B
    d.num_item = I
    d.ability_dim = spec.ability_dim
    d.irt_model = spec.irt_model
    d.posterior = (_lib.POSTERIOR_GIVEN if spec.given else
                   _lib.POSTERIOR_CONDITIONAL if spec.conditional else _lib.POSTERIOR_UNCONDITIONAL)
    d.missing_mode = _lib.MISSING_DROP if spec.drop_missing else _lib.MISSING_PRIOR
    d.mask_dtype = mask_code
    d.reg_mode = reg_mode
    d.n_flows = spec.n_flows
    d.want_grad = 1 if want_grad else 0
    d.deterministic = 1
    d.response_row_stride = resp_stride
    d.mask_row_stride = mask_stride
    return d


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('vibo_amd: the fused ELBO runs on the GPU only (tensor on %s); '
                               'there is no CPU path' % t.device)


ROW_COUNT_CACHE = True     # (tests / A-B runs switch it off)


def _resident_row_counts(spec, response, mask, mask_code):
    """Whole-row counts (vibo_row_counts) of a matrix this process keeps calling with -- or None.

    Rows of more than 1024 items under the unconditional posterior are counted in a pass of their own in front of the panels
    (half of the call: 5 B/cell).  The counts depend on the data alone, so a matrix seen a second time (same tensors, unchanged
    since: the resident training / evaluation split) is counted once, over all of its rows, and every later call -- the whole
    matrix or minibatches gathered from it through row_index -- hands them to vibo_elbo_fwd_bwd_counts.
    The record lives ON the response tensor (`_vibo_row_counts`): the counts are freed with the data they describe and never
    before -- a captured hipGraph that holds the data's address holds the counts' too."""
    I = response.shape[1]
    if (not ROW_COUNT_CACHE or spec.conditional or getattr(spec, 'given', False) or I <= 1024 or I > 32767
            or mask_code not in (_lib.MASK_NONE, _lib.MASK_U8, _lib.MASK_CODES)):
        return None
    mv = mask._version if mask is not None else 0
    rec = getattr(response, '_vibo_row_counts', None)      # [response version, weakref(mask) | None, mask version, counts | None]
    if rec is not None and rec[0] == response._version and (rec[1]() if rec[1] is not None else None) is mask and rec[2] == mv:
        if rec[3] is None:               # second sighting: count now
            if torch.cuda.is_current_stream_capturing():
                return None              # (a count recorded into a hipGraph would not have run when the next eager call reads it)
            lib = _lib.load()
            P = response.shape[0]
            d = _make_desc(spec, P, I, mask_code, _lib.REG_SAMPLED if spec.n_flows else _lib.REG_KL, False, response.stride(0),
                           mask.stride(0) if mask is not None else 0)
            counts = torch.empty(P, dtype=torch.int32, device=response.device)
            stream = ctypes.c_void_p(torch.cuda.current_stream(response.device).cuda_stream)
            _lib.check(lib.vibo_row_counts(ctypes.byref(d), _ptr(response), _ptr(mask), ctypes.c_void_p(0), _ptr(counts), stream), 'vibo_row_counts')
            rec[3] = counts
        return rec[3]
    response._vibo_row_counts = [response._version, weakref.ref(mask) if mask is not None else None, mv, None]
    return None


def _hip_launch_elbo(spec, response, mask, mask_code, row_index, table, item, eps, flow, reg_mode,
                     want_grad, num_person, train_step=None):
    """Single call into vibo_elbo_fwd_bwd on the current stream.
    train_step = (step_count tensor, skip_finalize): vibo_elbo_fwd_bwd_step instead -- the same call that also ticks Adam's
    step counter (the folded train step, trainer.FusedTrainer); with skip_finalize the partial records stay in
    RawElbo.workspace for vibo_train_epilogue_fused and `flat` is filled by that call."""
    lib = _lib.load()
    _require_device(response, mask, table, item, eps)
    dev = response.device
    I = response.shape[1]
    B = int(num_person)
    A, D = spec.ability_dim, spec.item_dim
    table_shape = tuple(table.shape) if table is not None else tuple(spec.table_shape(I))
    n_table = 1
    for n in table_shape:
        n_table *= n
    n_flow = spec.n_flows * (2 * A + 1)
    n_item = I * D
    flat = torch.empty(_lib.NUM_SCALARS + 2 * n_table + n_item + 2 * n_flow, dtype=torch.float32, device=dev)
    post = torch.empty(3, B, A, dtype=torch.float32, device=dev)
    ability_k = torch.empty(B, A, dtype=torch.float32, device=dev) if spec.n_flows else None
    ladj = torch.empty(B, dtype=torch.float32, device=dev) if spec.n_flows else None
    d = _make_desc(spec, B, I, mask_code, reg_mode, want_grad, response.stride(0),
                   mask.stride(0) if mask is not None else 0)
    ws_bytes = lib.vibo_workspace_bytes(ctypes.byref(d))
    if ws_bytes == 0:
        _lib.check(-1, 'vibo_workspace_bytes')
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    fbase, esz = flat.data_ptr(), 4
    o_tab = _lib.NUM_SCALARS
    o_item = o_tab + 2 * n_table
    o_flow = o_item + n_item
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if train_step is not None:
        rc = lib.vibo_elbo_fwd_bwd_step(
            ctypes.byref(d), _ptr(train_step[0]), 1 if train_step[1] else 0, _ptr(response), _ptr(mask), _ptr(row_index),
            _ptr(table), _ptr(item), _ptr(eps),
            ctypes.c_void_p(fbase), _ptr(post[0]), _ptr(post[1]), _ptr(post[2]),
            ctypes.c_void_p(fbase + esz * o_tab), ctypes.c_void_p(fbase + esz * o_item),
            _ptr(ws), ctypes.c_size_t(ws_bytes), stream)
        _lib.check(rc, 'vibo_elbo_fwd_bwd_step')
    elif (counts := _resident_row_counts(spec, response, mask, mask_code)) is not None and (row_index is not None or B == response.shape[0]):
        rc = lib.vibo_elbo_fwd_bwd_counts(
            ctypes.byref(d), _ptr(response), _ptr(mask), _ptr(row_index), _ptr(counts), _ptr(table), _ptr(item), _ptr(eps),
            _ptr(flow),
            ctypes.c_void_p(fbase), _ptr(post[0]), _ptr(post[1]), _ptr(post[2]), _ptr(ability_k), _ptr(ladj),
            ctypes.c_void_p(fbase + esz * o_tab), ctypes.c_void_p(fbase + esz * o_item),
            ctypes.c_void_p(fbase + esz * o_flow) if n_flow else ctypes.c_void_p(0),
            _ptr(ws), ctypes.c_size_t(ws_bytes), stream)
        _lib.check(rc, 'vibo_elbo_fwd_bwd_counts')
    else:
        rc = lib.vibo_elbo_fwd_bwd(
            ctypes.byref(d), _ptr(response), _ptr(mask), _ptr(row_index), _ptr(table), _ptr(item), _ptr(eps),
            _ptr(flow),
            ctypes.c_void_p(fbase), _ptr(post[0]), _ptr(post[1]), _ptr(post[2]), _ptr(ability_k), _ptr(ladj),
            ctypes.c_void_p(fbase + esz * o_tab), ctypes.c_void_p(fbase + esz * o_item),
            ctypes.c_void_p(fbase + esz * o_flow) if n_flow else ctypes.c_void_p(0),
            _ptr(ws), ctypes.c_size_t(ws_bytes), stream)
        _lib.check(rc, 'vibo_elbo_fwd_bwd')
    raw = RawElbo(flat=flat, n_table=n_table, n_item=n_item, n_flow=n_flow, table_shape=table_shape,
                  ability_mu=post[0], ability_logvar=post[1], ability=post[2],
                  ability_k=ability_k, ability_ladj=ladj)
    if train_step is not None and train_step[1]:
        raw.workspace = ws          # (kept alive until the epilogue has summed the partial records)
    return raw


def _hip_multi_forward(spec, response, mask, mask_code, row_index, table, items, eps, flow, reg_mode, num_person):
    """vibo_elbo_multi_forward: S forward evaluations in one pass.  items [S,I,D], eps [S,B,A] -> scalars [S,8], or
    None when the configuration is not on the row-split path (the caller then loops over single launches)."""
    lib = _lib.load()
    _require_device(response, mask, table, items, eps)
    S, B, I = int(items.shape[0]), int(num_person), response.shape[1]
    d = _make_desc(spec, B, I, mask_code, reg_mode, False, response.stride(0), mask.stride(0) if mask is not None else 0)
    ws_bytes = lib.vibo_multi_workspace_bytes(ctypes.byref(d), S)
    if ws_bytes == 0:
        return None
    dev = response.device
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.empty(S, _lib.NUM_SCALARS, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.vibo_elbo_multi_forward(ctypes.byref(d), S, _ptr(response), _ptr(mask), _ptr(row_index), _ptr(table),
                                     _ptr(items), _ptr(eps), _ptr(flow), _ptr(out), _ptr(ws), ctypes.c_size_t(ws_bytes),
                                     stream)
    if rc == -8:
        return None
    _lib.check(rc, 'vibo_elbo_multi_forward')
    return out


def _hip_encode(spec, response, mask, mask_code, row_index, table, num_person):
    lib = _lib.load()
    _require_device(response, mask, table)
    dev = response.device
    B, I, A = int(num_person), response.shape[1], spec.ability_dim
    out = torch.empty(2, B, A, dtype=torch.float32, device=dev)
    # (the posterior does not depend on the flows; the descriptor only has to be self-consistent)
    d = _make_desc(spec, B, I, mask_code, _lib.REG_SAMPLED if spec.n_flows else _lib.REG_KL, False, response.stride(0),
                   mask.stride(0) if mask is not None else 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ws_bytes = lib.vibo_workspace_bytes(ctypes.byref(d))          # scratch for the row statistics of the fast path
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
    rc = lib.vibo_encode(ctypes.byref(d), _ptr(response), _ptr(mask), _ptr(row_index), _ptr(table),
                         _ptr(out[0]), _ptr(out[1]), _ptr(ws), ctypes.c_size_t(ws_bytes), stream)
    _lib.check(rc, 'vibo_encode')
    return out[0], out[1]


def _hip_decode(spec, ability, item):
    lib = _lib.load()
    _require_device(ability, item)
    B, I = ability.shape[0], item.shape[0]
    out = torch.empty(B, I, dtype=torch.float32, device=ability.device)
    d = _make_desc(spec, B, I, _lib.MASK_NONE, _lib.REG_SAMPLED if spec.n_flows else _lib.REG_KL, False, I, 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream(ability.device).cuda_stream)
    rc = lib.vibo_decode(ctypes.byref(d), _ptr(ability), _ptr(item), _ptr(out), stream)
    _lib.check(rc, 'vibo_decode')
    return out


def _hip_row_counts(response, mask, mask_code, row_index):
    """vibo_row_counts: int32 [B], n_correct << 16 | n_observed per person row."""
    lib = _lib.load()
    _require_device(response, mask)
    B = int(row_index.numel()) if row_index is not None else response.shape[0]
    I = response.shape[1]
    out = torch.empty(B, dtype=torch.int32, device=response.device)
    d = _make_desc(ElboSpec(irt_model=1, ability_dim=1), B, I, mask_code, _lib.REG_KL, False, response.stride(0),
                   mask.stride(0) if mask is not None else 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream(response.device).cuda_stream)
    rc = lib.vibo_row_counts(ctypes.byref(d), _ptr(response), _ptr(mask), _ptr(row_index), _ptr(out), stream)
    _lib.check(rc, 'vibo_row_counts')
    return out


_COUNTS_CACHE = []         # [(weakref response, version, weakref mask | None, version, counts)], newest last, at most 4
_COUNTS_SEEN = []          # [weakref to the last response tensor a gathered minibatch came from]


def row_counts(response, mask, row_index=None):
    """int32 [B]: n_correct << 16 | n_observed per person -- the sufficient statistics of a Bernoulli response row for
    the unconditional encoders (here: the masked mean of --ability-merge mean, models.py:631-650).  The counts of a whole
    matrix are kept while the same (unmodified) tensors come back -- a resident dataset is counted once, not per step;
    minibatches gathered through `row_index` index into them."""
    key_r = response.codes if isinstance(response, CellCodes) else response       # the caller's own tensor objects
    key_m = mask
    for rr, rv, mr, mv, cnt in _COUNTS_CACHE:
        if rr() is key_r and rv == key_r._version and (
                (mr is None and key_m is None) or (mr is not None and mr() is key_m and mv == key_m._version)):
            return cnt if row_index is None else cnt[row_index]
    response, mask, code = prepare_rows(response, mask, keep_int64=True)
    if row_index is not None and not (_COUNTS_SEEN and _COUNTS_SEEN[0]() is key_r):
        # a gathered minibatch of a matrix seen for the first time: count those rows only; the whole matrix is counted
        # (once) when the same tensor comes back
        _COUNTS_SEEN[:] = [weakref.ref(key_r)]
        return _BACKEND['counts'](response, mask, code, row_index)
    cnt = _BACKEND['counts'](response, mask, code, None)
    try:
        key_r._vibo_counts_keepalive = cnt      # (alive as long as the data: a captured hipGraph that reads the data reads these too)
    except AttributeError:
        pass
    _COUNTS_CACHE[:] = [e for e in _COUNTS_CACHE if e[0]() is not None and (e[2] is None or e[2]() is not None)][-3:]
    _COUNTS_CACHE.append((weakref.ref(key_r), key_r._version, weakref.ref(key_m) if key_m is not None else None,
                          key_m._version if key_m is not None else 0, cnt))
    return cnt if row_index is None else cnt[row_index]


def _mean_desc(counts, A):
    return _make_desc(ElboSpec(irt_model=1, ability_dim=A), int(counts.numel()), 1, _lib.MASK_NONE, _lib.REG_KL, False, 1, 0)


def _hip_mean_encoder_fwd(counts, u, v, w2, b2):
    lib = _lib.load()
    _require_device(counts, u, v, w2, b2)
    A2, H = w2.shape
    post = torch.empty(counts.numel(), A2, dtype=torch.float32, device=counts.device)
    d = _mean_desc(counts, A2 // 2)
    stream = ctypes.c_void_p(torch.cuda.current_stream(counts.device).cuda_stream)
    rc = lib.vibo_mean_encoder_forward(ctypes.byref(d), H, _ptr(counts), _ptr(u), _ptr(v), _ptr(w2), _ptr(b2), _ptr(post), stream)
    _lib.check(rc, 'vibo_mean_encoder_forward')
    return post


def _hip_mean_encoder_bwd(counts, u, v, w2, gpost):
    """-> (d/du [H], d/dv [H], d/dW2 [2A,H], d/db2 [2A]) summed over the persons."""
    lib = _lib.load()
    _require_device(counts, u, v, w2, gpost)
    A2, H = w2.shape
    d = _mean_desc(counts, A2 // 2)
    n_part = lib.vibo_mean_encoder_partials(ctypes.byref(d))
    part = torch.empty(n_part, 2 * H + A2 * H + A2, dtype=torch.float32, device=counts.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(counts.device).cuda_stream)
    rc = lib.vibo_mean_encoder_backward(ctypes.byref(d), H, _ptr(counts), _ptr(u), _ptr(v), _ptr(w2), _ptr(gpost), _ptr(part),
                                        n_part, stream)
    _lib.check(rc, 'vibo_mean_encoder_backward')
    tot = part.sum(0)
    return tot[:H], tot[H:2 * H], tot[2 * H:2 * H + A2 * H].view(A2, H), tot[2 * H + A2 * H:]


class MeanEncoderFn(torch.autograd.Function):
    """(u, v, W2, b2) -> posterior [B, 2A] = W2 elu(u + w_p v) + b2 with w_p from the packed row counts
    (vibo_mean_encoder_forward / _backward; see include/vibo_hip.h)."""

    @staticmethod
    def forward(ctx, u, v, w2, b2, counts):
        u, v, w2, b2 = (t.detach().contiguous().float() for t in (u, v, w2, b2))
        ctx.save_for_backward(u, v, w2, counts)
        return _BACKEND['mean_fwd'](counts, u, v, w2, b2)

    @staticmethod
    def backward(ctx, g):
        u, v, w2, counts = ctx.saved_tensors
        gu, gv, gw2, gb2 = _BACKEND['mean_bwd'](counts, u, v, w2, g.contiguous().float())
        return gu, gv, gw2, gb2, None


class FlowStackFn(torch.autograd.Function):
    """(z [N, D], packed [K, 2D+1] = uhat | w | b) -> (z_K [N, D], ladj [N]): a stack of planar flows in two launches
    (vibo_flow_stack_forward / _backward) instead of ~35 small PyTorch kernels per flow and direction."""

    @staticmethod
    def forward(ctx, z, packed):
        lib = _lib.load()
        z, packed = z.detach().contiguous().float(), packed.detach().contiguous().float()
        _require_device(z, packed)
        N, D = z.shape
        K = packed.shape[0]
        z_out, ladj, th = torch.empty_like(z), torch.empty(N, dtype=torch.float32, device=z.device), torch.empty(N, K, dtype=torch.float32, device=z.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
        _lib.check(lib.vibo_flow_stack_forward(N, D, K, _ptr(z), _ptr(packed), _ptr(z_out), _ptr(ladj), _ptr(th), stream),
                   'vibo_flow_stack_forward')
        ctx.save_for_backward(z_out, packed, th)
        return z_out, ladj

    @staticmethod
    def backward(ctx, g_z, g_l):
        lib = _lib.load()
        z_out, packed, th = ctx.saved_tensors
        N, D = z_out.shape
        K = packed.shape[0]
        g_z = torch.zeros_like(z_out) if g_z is None else g_z.contiguous().float()
        g_l = torch.zeros(N, dtype=torch.float32, device=z_out.device) if g_l is None else g_l.contiguous().float()
        g_in = torch.empty_like(z_out)
        parts = torch.empty((N + 255) // 256, K, 2 * D + 1, dtype=torch.float32, device=z_out.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(z_out.device).cuda_stream)
        _lib.check(lib.vibo_flow_stack_backward(N, D, K, _ptr(z_out), _ptr(packed), _ptr(th), _ptr(g_z), _ptr(g_l), _ptr(g_in),
                                                _ptr(parts), stream), 'vibo_flow_stack_backward')
        return g_in, parts.sum(0)


def _hip_flow_stack(z, packed):
    return FlowStackFn.apply(z, packed)


def _hip_decode_mean(spec, abilities, items):
    """vibo_decode_mean: abilities [S,B,A], items [S,I,D] -> mean over S of P(response = 1) [B,I]."""
    lib = _lib.load()
    _require_device(abilities, items)
    S, B, I = int(abilities.shape[0]), int(abilities.shape[1]), int(items.shape[1])
    out = torch.empty(B, I, dtype=torch.float32, device=abilities.device)
    d = _make_desc(spec, B, I, _lib.MASK_NONE, _lib.REG_SAMPLED if spec.n_flows else _lib.REG_KL, False, I, 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream(abilities.device).cuda_stream)
    rc = lib.vibo_decode_mean(ctypes.byref(d), S, _ptr(abilities), _ptr(items), _ptr(out), stream)
    _lib.check(rc, 'vibo_decode_mean')
    return out


class CodeTableSumFn(torch.autograd.Function):
    """S[p, :] = sum over person p's observed cells of feature[code_pi, i, :]  (vibo_code_table_sum_forward / _backward: the
    one-hot [B, 2I] x [2I, H] contraction of --ability-merge mean with --conditional-posterior and its transpose, on the
    matrix pipe from the 1-byte cell codes).  feature [2, I, 64] fp32; codes = CellCodes of the minibatch's rows."""

    @staticmethod
    def forward(ctx, feature, cell_codes):
        lib = _lib.load()
        codes = cell_codes.codes
        B, I = codes.shape
        H = feature.shape[2]
        feat = feature.detach().contiguous().float()
        _require_device(feat, codes)
        out = torch.empty(B, H, dtype=torch.float32, device=codes.device)
        nbytes = lib.vibo_code_table_scratch_bytes(B, I, H)
        if nbytes == 0:
            raise _lib.ViboLibraryError('vibo_code_table_sum: hidden_dim must be 64')
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=codes.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(codes.device).cuda_stream)
        rc = lib.vibo_code_table_sum_forward(B, I, H, _ptr(codes), codes.stride(0), _ptr(feat), _ptr(out), _ptr(scratch), nbytes, stream)
        _lib.check(rc, 'vibo_code_table_sum_forward')
        ctx.cell_codes, ctx.shape = cell_codes, (I, H)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        codes = ctx.cell_codes.codes
        B = codes.shape[0]
        I, H = ctx.shape
        g = g.contiguous().float()
        dfeat = torch.empty(2, I, H, dtype=torch.float32, device=codes.device)
        nbytes = lib.vibo_code_table_scratch_bytes(B, I, H)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=codes.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(codes.device).cuda_stream)
        rc = lib.vibo_code_table_sum_backward(B, I, H, _ptr(codes), codes.stride(0), _ptr(g), _ptr(dfeat), _ptr(scratch), nbytes, stream)
        _lib.check(rc, 'vibo_code_table_sum_backward')
        return dfeat, None


def _hip_cond_mean_sum(feature, response, mask, row_index):
    """(S [B, H], n_observed [B]) for the minibatch: the cells as CellCodes (packed here from fp32 rows + mask if need be),
    the sum on the matrix pipe (CodeTableSumFn), the observed counts from vibo_row_counts."""
    if isinstance(response, CellCodes):
        cc = response.rows(row_index) if row_index is not None else response
    else:
        r = prepare_response(response)
        m = None if mask is None else prepare_mask(mask)[0]
        if row_index is not None:
            r, m = r[row_index], (None if m is None else m[row_index])
        cc = pack_cell_codes(r, m)
    if cc.codes.shape[1] > 32767:
        # (vibo_row_counts packs n_correct << 16 | n_observed into an int32: wider rows are counted by torch -- one more pass over
        #  the codes, on a width the reference itself never trains at)
        nobs = (cc.codes != 2).sum(dim=1).to(feature.dtype)
    else:
        counts = _BACKEND['counts'](cc.codes, cc.codes, _lib.MASK_CODES, None)
        nobs = (counts & 0xffff).to(feature.dtype)
    return CodeTableSumFn.apply(feature, cc), nobs


# The entry points of the native library.  tests/ swap these for the CPU
# oracle to exercise the host logic without a GPU (never done by product code).
_BACKEND = {'elbo': _hip_launch_elbo, 'encode': _hip_encode, 'decode': _hip_decode, 'multi': _hip_multi_forward,
            'decode_mean': _hip_decode_mean, 'counts': _hip_row_counts, 'mean_fwd': _hip_mean_encoder_fwd,
            'mean_bwd': _hip_mean_encoder_bwd, 'flow_stack': _hip_flow_stack, 'cond_mean_sum': _hip_cond_mean_sum}


class FusedELBO(torch.autograd.Function):
    """(table, item, flow) -> (LL, REG, scalars, ability_mu, ability_logvar,
    ability, ability_k, ability_ladj).

    LL  = sum of masked Bernoulli log-likelihoods (utils.py:46-49, models.py:399)
    REG = sum_p KL(q(theta_p)||N(0,1))  (reg_mode KL,  models.py:428) or
          sum_p [log q(theta_p) - log p(theta_p)] at the sample (SAMPLED, models.py:412-418,433-435)
    The kernel has already produced d LL/d(.) and d REG/d(.); backward only scales
    and adds them, so loss.backward() costs no second pass over the responses.
    """

    @staticmethod
    def forward(ctx, table, item, flow, response, mask, mask_code, row_index, eps, spec, reg_mode,
                num_person, reducer):
        need_grad = any(ctx.needs_input_grad[:3])
        table_c = table.detach().contiguous()
        item_c = item.detach().contiguous()
        flow_c = flow.detach().contiguous() if flow is not None else None
        raw = _BACKEND['elbo'](spec, response, mask, mask_code, row_index, table_c, item_c,
                               eps.contiguous(), flow_c, reg_mode, need_grad, num_person)
        if reducer is not None:      # person-sharded data parallelism: ONE all-reduce (RCCL) per step
            if spec.given:           # (the per-person posterior gradients in the middle of the buffer stay local)
                reducer(raw.flat[:_lib.NUM_SCALARS])
                if raw.n_item + 2 * raw.n_flow:
                    reducer(raw.flat[_lib.NUM_SCALARS + 2 * raw.n_table:])
            else:
                reducer(raw.flat)
        ctx.raw = raw
        ctx.item_shape = tuple(item.shape)
        ctx.has_flow = flow is not None
        ctx.need_grad = need_grad
        sc = raw.scalars
        outs = (sc[_lib.S_LL].clone(), sc[_lib.S_REG].clone(), sc.clone(),
                raw.ability_mu, raw.ability_logvar, raw.ability,
                raw.ability_k if raw.ability_k is not None else raw.ability,
                raw.ability_ladj if raw.ability_ladj is not None else sc.new_zeros(()))
        ctx.mark_non_differentiable(*outs[2:])
        return outs

    @staticmethod
    def backward(ctx, g_ll, g_reg, *_):
        raw = ctx.raw
        if not ctx.need_grad:
            return (None,) * 12
        g_table = g_item = g_flow = None
        if ctx.needs_input_grad[0]:
            g_table = g_ll * raw.grad_table(0) + g_reg * raw.grad_table(1)
        if ctx.needs_input_grad[1]:
            g_item = g_ll * raw.grad_item(ctx.item_shape)
        if ctx.has_flow and ctx.needs_input_grad[2]:
            g_flow = (g_ll * raw.grad_flow(0) + g_reg * raw.grad_flow(1)).view(-1, 2 * raw.ability_mu.shape[1] + 1)
        return (g_table, g_item, g_flow) + (None,) * 9


def fused_elbo(spec, table, item, flow, response, mask, eps, *, reg_mode=_lib.REG_KL, row_index=None,
               reducer=None):
    """Functional front end.  response [B,I(,1)] fp32, mask [B,I(,1)] bool/int64/None,
    table per ElboSpec.table_shape, item [I,D], eps [B,A], flow [n_flows,2A+1] or None.
    With row_index (int64 [B]) the rows response[row_index] / mask[row_index] are
    gathered inside the kernel (device-resident dataset, shuffled minibatches)."""
    response, mask, code = prepare_rows(response, mask)
    I = response.shape[1]
    spec.check_supported(I)
    B = int(row_index.numel()) if row_index is not None else response.shape[0]
    if tuple(table.shape) != spec.table_shape(I, B):
        raise ValueError(f'table shape {tuple(table.shape)} != {spec.table_shape(I, B)}')
    if tuple(item.shape) != (I, spec.item_dim):
        raise ValueError(f'item shape {tuple(item.shape)} != {(I, spec.item_dim)}')
    if tuple(eps.shape) != (B, spec.ability_dim):
        raise ValueError(f'eps shape {tuple(eps.shape)} != {(B, spec.ability_dim)}')
    return FusedELBO.apply(table, item, flow, response, mask, code, row_index, eps, spec, reg_mode, B, reducer)


def encode_posterior(spec, table, response, mask, row_index=None):
    """Forward-only q(ability | responses): (mu, logvar) [B,A] (models.py:356-371 under no_grad)."""
    response, mask, code = prepare_rows(response, mask)
    B = int(row_index.numel()) if row_index is not None else response.shape[0]
    return _BACKEND['encode'](spec, response, mask, code, row_index, table.detach().contiguous(), B)


def decode_probs_mean(spec, abilities, items):
    """Mean over S posterior draws of P(response = 1): abilities [S,B,A], items [S,I,D] -> [B,I]
    (vibo.py:363-390 reduced as at :504-548, without the [S,B,I] stack)."""
    return _BACKEND['decode_mean'](spec, abilities.detach().contiguous().float(), items.detach().contiguous().float())


def decode_probs(spec, ability, item):
    """P(response = 1) [B,I] (models.py:729-766)."""
    return _BACKEND['decode'](spec, ability.detach().contiguous().float(), item.detach().contiguous().float())
