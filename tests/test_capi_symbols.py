"""The C-ABI library loads and exports every symbol include/vibo_hip.h declares
(no compute: runs without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from vibo_amd import _lib


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'vibo_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vibo_[a-z_0-9]+)\s*\(', src)))


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), 'run python __graft_entry__.py (build()) first'


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert set(names) == set(_lib.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n


def test_version_and_descriptor_validation():
    lib = _lib.load()
    assert lib.vibo_version() == _lib.ABI_VERSION
    d = _lib.ViboDesc()
    assert lib.vibo_workspace_bytes(ctypes.byref(d)) == 0          # abi_version 0 -> rejected
    assert b'abi_version' in lib.vibo_last_error_string()
    d.abi_version = _lib.ABI_VERSION
    d.num_person, d.num_item, d.ability_dim, d.irt_model = 128, 100, 17, 2
    assert lib.vibo_workspace_bytes(ctypes.byref(d)) == 0          # ability_dim 17 > VIBO_MAX_ABILITY_DIM_WIDE
    d.ability_dim = 8
    assert lib.vibo_workspace_bytes(ctypes.byref(d)) > 0
    d.ability_dim = 12                                              # 9..16: the wave-per-person kernel, whatever the shape
    assert lib.vibo_workspace_bytes(ctypes.byref(d)) > 0 and lib.vibo_plan_kernel(ctypes.byref(d)) == 5
    d.posterior = _lib.POSTERIOR_GIVEN                              # ... but not a caller-supplied posterior, nor cell codes
    assert lib.vibo_workspace_bytes(ctypes.byref(d)) == 0 and b'row-split' in lib.vibo_last_error_string()
    assert _lib.MAX_ABILITY_DIM == 16 and _lib.MAX_ABILITY_DIM_FAST == 8


def test_planner_reports_its_kernel_and_honours_the_flags():
    """vibo_plan_kernel (no launch, no GPU needed): the benchmark's shape goes to the matrix row-split kernel, the
    reference's default minibatch (16 persons) to the VALU one, and vibo_desc.flags pins either -- the planner reads the
    descriptor only (no environment variables)."""
    lib = _lib.load()

    def plan(B, I, A, flags=0, mask=_lib.MASK_U8):
        d = _lib.ViboDesc()
        d.abi_version = _lib.ABI_VERSION
        d.num_person, d.num_item, d.ability_dim, d.irt_model = B, I, A, 2
        d.mask_dtype, d.want_grad, d.flags = mask, 1, flags
        d.response_row_stride = d.mask_row_stride = (I + 3) & ~3
        return lib.vibo_plan_kernel(ctypes.byref(d))

    assert plan(1_000_000, 1000, 8) == 1 and plan(1_000_000, 1000, 1) == 1          # bench.py's two shapes
    assert plan(4096, 1000, 8) == 1                                                  # ... and its elbo_rel_err sample
    assert plan(16, 1000, 8) == 2
    # (round 6 calibration: the matrix kernel from 2 048 persons at 896+ items and 5+ dims; no 385..512-item exclusion any more)
    assert plan(2048, 1000, 8) == 1 and plan(2048, 768, 8) == 2 and plan(4096, 512, 8) == 1 and plan(4096, 256, 8) == 2
    # narrow rows of the plain model (BASELINE configs[0] / [3]: 100 / 95 items) go to the narrow-row kernel at any minibatch size,
    # unless a flag pins a row-split kernel or the posterior is wider than 4 dims
    assert plan(535_598, 96, 1) == 6 and plan(16, 100, 1) == 6 and plan(8000, 95, 4) == 6 and plan(1000, 128, 2) == 6
    assert plan(535_598, 96, 1, _lib.FLAG_KERNEL_VALU) == 2 and plan(535_598, 96, 1, _lib.FLAG_KERNEL_MATRIX) == 1
    assert plan(535_598, 96, 8) == 2 and plan(535_598, 132, 1) == 2
    assert plan(16, 1000, 8, _lib.FLAG_KERNEL_MATRIX) == 1 and plan(1_000_000, 1000, 8, _lib.FLAG_KERNEL_VALU) == 2
    assert plan(1000, 1000, 2, 0, _lib.MASK_I64) == 3
    assert plan(16, 1000, 8, 3) < 0 and plan(16, 1000, 8, 64) < 0                    # contradictory / unknown flags
    # the conditional posterior's two extra passes on the matrix pipe when the rows are (or become) cell codes, from a size that
    # depends on ability_dim (measured table in make_plan); fp32 rows keep the VALU pre pass (it emits the codes) up to 4 dims
    def cond(B, I, A, flags=0, mask=_lib.MASK_U8, grad=1):
        d = _lib.ViboDesc()
        d.abi_version = _lib.ABI_VERSION
        d.num_person, d.num_item, d.ability_dim, d.irt_model = B, I, A, 2
        d.posterior = _lib.POSTERIOR_CONDITIONAL
        d.mask_dtype, d.want_grad, d.flags = mask, grad, flags
        d.response_row_stride = d.mask_row_stride = (I + 3) & ~3
        return lib.vibo_plan_cond_passes(ctypes.byref(d))

    assert cond(1_000_000, 1000, 8, mask=_lib.MASK_CODES) == 3 and cond(1_000_000, 1000, 1, mask=_lib.MASK_CODES) == 3
    assert cond(1_000_000, 1000, 8) == 3 and cond(1_000_000, 1000, 4) == 2
    # ability_dim 1 on fp32 rows: no first pass at all where the matrix kernel runs (bit 2: it forms the experts' sums itself) -- from
    # 8 192 persons at 256..1024 items (4 096 at 896+); the three-pass pin and cell-code rows keep the separate pass
    assert cond(1_000_000, 1000, 1) == 6 and cond(8192, 1000, 1) == 6 and cond(8192, 256, 1) == 6 and cond(4096, 1000, 1) == 6 and cond(4096, 768, 1) == 2 and cond(2048, 1000, 1) == 2
    assert cond(1_000_000, 1000, 1, _lib.FLAG_COND_THREE_PASS) == 2 and cond(1_000_000, 1000, 1, grad=0) == 4
    assert cond(1_000_000, 1000, 2) == 2 and cond(1_000_000, 1000, 1, _lib.FLAG_NO_EMIT_CODES) == 0
    assert cond(16, 1000, 8, mask=_lib.MASK_CODES) == 3 and cond(16, 1000, 8) == 3       # 5+ dims: at any size
    assert cond(16, 1000, 1, mask=_lib.MASK_CODES) == 0 and cond(4095, 1000, 1, mask=_lib.MASK_CODES) == 0
    assert cond(16, 1000, 1, _lib.FLAG_COND_MATRIX, _lib.MASK_CODES) == 3 and cond(16, 1000, 1, _lib.FLAG_COND_MATRIX) == 2
    assert cond(16, 1000, 3) == 2 and cond(1024, 1000, 1) == 0 and cond(2048, 1000, 1) == 2
    assert cond(16, 1000, 8, _lib.FLAG_COND_MATRIX | _lib.FLAG_COND_VALU) < 0
    assert cond(1_000_000, 1000, 8, _lib.FLAG_COND_VALU, _lib.MASK_CODES) == 0
    assert cond(1_000_000, 1000, 8, _lib.FLAG_NO_EMIT_CODES) == 0                    # fp32 rows that stay fp32: nothing to read
    assert cond(1_000_000, 1000, 8, mask=_lib.MASK_CODES, grad=0) == 1               # forward only: no gradient pass at all
    assert plan(1_000_000, 1000, 8) == 1 and lib.vibo_plan_cond_passes is not None
    import os
    os.environ['VIBO_MSPLIT'] = '0'                                                  # (round 2's switch: ignored now)
    try:
        assert plan(1_000_000, 1000, 8) == 1
    finally:
        del os.environ['VIBO_MSPLIT']


def test_desc_struct_matches_header():
    """ctypes mirror has the header's field order."""
    src = open(os.path.join(ROOT, 'include', 'vibo_hip.h')).read()
    body = src[src.index('typedef struct vibo_desc {'):src.index('} vibo_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = re.findall(r'int(?:32|64)_t\s+([a-z_]+)\s*;', body)
    assert fields == [f[0] for f in _lib.ViboDesc._fields_]


def test_folded_step_entry_points_check_their_arguments():
    """The host-side checks of the folded-step entry points answer without a GPU."""
    lib = _lib.load()
    d = _lib.ViboDesc()
    assert lib.vibo_train_step_supported(ctypes.byref(d)) == 0                  # (zeroed descriptor: wrong abi_version)
    assert lib.vibo_elbo_fwd_bwd_step(ctypes.byref(d), None, 0, *([None] * 13), 0, None) == -5
    assert lib.vibo_train_prime(ctypes.byref(d), 64, *([None] * 10)) == -5


def test_decoder_desc_struct_matches_header():
    src = open(os.path.join(ROOT, 'include', 'vibo_hip.h')).read()
    body = src[src.index('typedef struct vibo_decoder_desc {'):src.index('} vibo_decoder_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for typ, names in re.findall(r'(int32_t|int64_t|float)\s+([a-z_, ]+);', body):
        fields += [(n.strip(), typ) for n in names.split(',')]
    ctype = {'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float}
    assert [(n, ctype[t]) for n, t in fields] == list(_lib.ViboDecoderDesc._fields_)
    lib = _lib.load()
    assert lib.vibo_decoder_person_chunks(0, 10) == 0 and lib.vibo_decoder_person_chunks(1000, 100) >= 1
    d = _lib.ViboDecoderDesc(16, 100, 32, 0, 1, 0.0, 100, 100)          # hidden_dim 32: refused before any launch
    assert lib.vibo_decoder_fwd_bwd(ctypes.byref(d), *([None] * 20)) == -6


def test_planner_thresholds_agree_with_the_committed_calibration_tables():
    """want_msplit's thresholds (csrc/vibo_capi.hip) are hand-written from tools/calibrate_planner.py's tables; tools/check_planner_table.py
    replays the committed tables (two MI355X boxes, profiles/*planner_calibration*.txt) against vibo_plan_kernel: no calibrated
    shape may go to the kernel that was measured more than 15 % slower on every box (the one island the rules do not follow --
    16 384 x 256 at ability_dim 8, 11-12 % -- stays under that; the tool's default 8 % lists it)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import check_planner_table
    bad, n = check_planner_table.check(tol=0.15, verbose=False)
    assert n >= 70 and not bad, bad


def test_matrix_kernel_instantiations_carry_no_scratch():
    """Every instantiation of the matrix row-split kernel (and the narrow-row kernel) is built without spilled registers or private
    memory: a spill reload in the batch loop is a scratch load + s_waitcnt vmcnt(0) that also drains the row prefetch (DESIGN 8.2:
    it cost the wide 3PL + flows instantiation 30 %).  Read from the code-object notes of the in-tree objects (tools/kernel_regs.sh
    does the same by hand); skipped when the build directory or the LLVM tools are not there."""
    import glob
    import re
    import subprocess
    import tempfile
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    bdir = os.path.join(root, 'variational-item-response-theory-public_amd', 'csrc', 'build')
    llvm = '/opt/rocm/lib/llvm/bin'
    objs = sorted(glob.glob(os.path.join(bdir, 'vibo_msplit_*.o'))) + sorted(glob.glob(os.path.join(bdir, 'vibo_narrow.o')))
    if not objs or not os.path.exists(os.path.join(llvm, 'llvm-readelf')):
        pytest.skip('no in-tree objects / LLVM tools')
    seen = 0
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
            subprocess.run([os.path.join(llvm, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, o], check=True)
            subprocess.run([os.path.join(llvm, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                            '--input=' + fat, '--output=' + co, '--unbundle'], check=True)
            notes = subprocess.run([os.path.join(llvm, 'llvm-readelf'), '--notes', co], check=True, capture_output=True, text=True).stdout
            for blk in notes.split('  - .agpr_count:')[1:]:
                name = re.search(r'\.name:\s+(\S+)', blk).group(1)
                if 'msplit_kernel' not in name and 'narrow_kernel' not in name:
                    continue
                spill = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', blk).group(1))
                scratch = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1))
                assert spill == 0 and scratch == 0, (os.path.basename(o), name, spill, scratch)
                seen += 1
    assert seen >= 216
