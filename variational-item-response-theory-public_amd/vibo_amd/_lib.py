"""ctypes binding of libvibo_hip.so (C ABI declared in include/vibo_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the HIP library is
missing or a call fails, this module raises.
"""
import ctypes
import os

ABI_VERSION = 2
NUM_SCALARS = 8
S_LL, S_REG, S_KL, S_LOGQ0, S_LOGP, S_LADJ, S_NOBS = 0, 1, 2, 3, 4, 5, 6

IRT_1PL, IRT_2PL, IRT_3PL = 1, 2, 3
POSTERIOR_UNCONDITIONAL, POSTERIOR_CONDITIONAL, POSTERIOR_GIVEN = 0, 1, 2
MISSING_PRIOR, MISSING_DROP = 0, 1
MASK_U8, MASK_I64, MASK_NONE, MASK_CODES = 0, 1, 2, 3
REG_KL, REG_SAMPLED = 0, 1
FLAG_KERNEL_VALU, FLAG_KERNEL_MATRIX, FLAG_NO_EMIT_CODES, FLAG_COND_VALU, FLAG_COND_MATRIX, FLAG_COND_THREE_PASS = 1, 2, 4, 8, 16, 32
KERNEL_NAMES = {1: 'matrix row-split (msplit_kernel)', 2: 'VALU row-split (split_kernel)', 3: 'wave-per-row', 4: 'tiled',
                5: 'wave-per-person', 6: 'narrow rows (narrow_kernel)'}
MAX_ABILITY_DIM = 16          # vibo_elbo_fwd_bwd / vibo_encode / vibo_decode (9..16: the wave-per-person kernel)
MAX_ABILITY_DIM_FAST = 8      # row-split / matrix-pipe kernels, trainers' native conditional / flow step, mean merge
MAX_FLOWS = 8

LIB_NAME = 'libvibo_hip.so'
# VIBO_HIP_LIB points at another build of the same ABI (A/B timing of kernel variants); default: the in-tree build
LIB_PATH = os.environ.get('VIBO_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


class ViboDesc(ctypes.Structure):
    """struct vibo_desc (include/vibo_hip.h)."""
    _fields_ = [
        ('abi_version', ctypes.c_int32),
        ('num_person', ctypes.c_int32),
        ('num_item', ctypes.c_int32),
        ('ability_dim', ctypes.c_int32),
        ('irt_model', ctypes.c_int32),
        ('posterior', ctypes.c_int32),
        ('missing_mode', ctypes.c_int32),
        ('mask_dtype', ctypes.c_int32),
        ('reg_mode', ctypes.c_int32),
        ('n_flows', ctypes.c_int32),
        ('want_grad', ctypes.c_int32),
        ('deterministic', ctypes.c_int32),
        ('response_row_stride', ctypes.c_int64),
        ('mask_row_stride', ctypes.c_int64),
        ('flags', ctypes.c_int32),
        ('reserved', ctypes.c_int32),
    ]


class ViboDecoderDesc(ctypes.Structure):
    """struct vibo_decoder_desc (include/vibo_hip.h)."""
    _fields_ = [
        ('num_person', ctypes.c_int32),
        ('num_item', ctypes.c_int32),
        ('hidden_dim', ctypes.c_int32),
        ('want_grad', ctypes.c_int32),
        ('person_chunks', ctypes.c_int32),
        ('resid', ctypes.c_float),
        ('response_row_stride', ctypes.c_int64),
        ('mask_row_stride', ctypes.c_int64),
    ]


EXPORTED_SYMBOLS = ('vibo_version', 'vibo_last_error_string', 'vibo_workspace_bytes', 'vibo_plan_kernel', 'vibo_plan_cond_passes',
                    'vibo_elbo_fwd_bwd', 'vibo_encode', 'vibo_decode', 'vibo_train_prologue', 'vibo_train_epilogue', 'vibo_fill_normal', 'vibo_multi_workspace_bytes',
                    'vibo_elbo_multi_forward', 'vibo_decode_mean', 'vibo_pack_codes', 'vibo_row_counts', 'vibo_mean_encoder_partials',
                    'vibo_mean_encoder_forward', 'vibo_mean_encoder_backward', 'vibo_train_prologue_noise',
                    'vibo_decoder_person_chunks', 'vibo_decoder_fwd_bwd', 'vibo_flow_stack_forward', 'vibo_flow_stack_backward',
                    'vibo_ctrain_param_floats', 'vibo_ctrain_scratch_floats', 'vibo_ctrain_prologue', 'vibo_ctrain_epilogue',
                    'vibo_code_table_scratch_bytes', 'vibo_code_table_sum_forward', 'vibo_code_table_sum_backward',
                    'vibo_train_step_supported', 'vibo_elbo_fwd_bwd_step', 'vibo_train_epilogue_fused', 'vibo_train_prime',
                    'vibo_mtrain_param_floats', 'vibo_mtrain_prologue', 'vibo_mean_encoder_backward_sets', 'vibo_mtrain_epilogue',
                    'vibo_set_insitu_timer', 'vibo_insitu_timer_reset', 'vibo_selftest_lane_swaps', 'vibo_elbo_fwd_bwd_counts')

_lib = None


class ViboLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ViboLibraryError(
            f'{LIB_PATH} not found: build the HIP extension first '
            f'(python __graft_entry__.py, or make -C variational-item-response-theory-public_amd/csrc). '
            f'There is no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64p, fp = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p
    dp = ctypes.POINTER(ViboDesc)
    lib.vibo_version.restype = ctypes.c_int
    lib.vibo_version.argtypes = []
    lib.vibo_last_error_string.restype = ctypes.c_char_p
    lib.vibo_last_error_string.argtypes = []
    lib.vibo_workspace_bytes.restype = ctypes.c_size_t
    lib.vibo_workspace_bytes.argtypes = [dp]
    lib.vibo_plan_kernel.restype = ctypes.c_int
    lib.vibo_plan_kernel.argtypes = [dp]
    lib.vibo_plan_cond_passes.restype = ctypes.c_int
    lib.vibo_plan_cond_passes.argtypes = [dp]
    lib.vibo_elbo_fwd_bwd.restype = ctypes.c_int
    lib.vibo_elbo_fwd_bwd.argtypes = [dp, fp, vp, i64p, fp, fp, fp, fp,      # inputs
                                      fp, fp, fp, fp, fp, fp,                # scalars + posterior outputs
                                      fp, fp, fp,                            # grads
                                      vp, ctypes.c_size_t, vp]               # workspace, stream
    lib.vibo_elbo_fwd_bwd_counts.restype = ctypes.c_int
    lib.vibo_elbo_fwd_bwd_counts.argtypes = [dp, fp, vp, i64p, vp, fp, fp, fp, fp,
                                             fp, fp, fp, fp, fp, fp,
                                             fp, fp, fp,
                                             vp, ctypes.c_size_t, vp]
    lib.vibo_encode.restype = ctypes.c_int
    lib.vibo_encode.argtypes = [dp, fp, vp, i64p, fp, fp, fp, vp, ctypes.c_size_t, vp]
    lib.vibo_decode.restype = ctypes.c_int
    lib.vibo_decode.argtypes = [dp, fp, fp, fp, vp]
    lib.vibo_train_prologue.restype = ctypes.c_int
    lib.vibo_train_prologue.argtypes = [dp, ctypes.c_int] + [fp] * 8 + [vp, vp]
    lib.vibo_train_prologue_noise.restype = ctypes.c_int
    lib.vibo_train_prologue_noise.argtypes = [dp, ctypes.c_int] + [fp] * 8 + [vp, ctypes.c_uint64, fp, ctypes.c_uint32, vp]
    lib.vibo_train_epilogue.restype = ctypes.c_int
    lib.vibo_train_epilogue.argtypes = [dp, ctypes.c_int] + [fp] * 6 + [vp] + [fp] * 8 + [vp]
    lib.vibo_fill_normal.restype = ctypes.c_int
    lib.vibo_fill_normal.argtypes = [fp, ctypes.c_int64, ctypes.c_uint64, vp, ctypes.c_uint32, vp]
    lib.vibo_multi_workspace_bytes.restype = ctypes.c_size_t
    lib.vibo_multi_workspace_bytes.argtypes = [dp, ctypes.c_int]
    lib.vibo_elbo_multi_forward.restype = ctypes.c_int
    lib.vibo_elbo_multi_forward.argtypes = [dp, ctypes.c_int, fp, vp, i64p, fp, fp, fp, fp, fp, vp, ctypes.c_size_t, vp]
    lib.vibo_row_counts.restype = ctypes.c_int
    lib.vibo_row_counts.argtypes = [dp, fp, vp, i64p, vp, vp]
    lib.vibo_mean_encoder_partials.restype = ctypes.c_int
    lib.vibo_mean_encoder_partials.argtypes = [dp]
    lib.vibo_mean_encoder_forward.restype = ctypes.c_int
    lib.vibo_mean_encoder_forward.argtypes = [dp, ctypes.c_int, vp, fp, fp, fp, fp, fp, vp]
    lib.vibo_mean_encoder_backward.restype = ctypes.c_int
    lib.vibo_mean_encoder_backward.argtypes = [dp, ctypes.c_int, vp, fp, fp, fp, fp, fp, ctypes.c_int, vp]
    lib.vibo_pack_codes.restype = ctypes.c_int
    lib.vibo_pack_codes.argtypes = [dp, fp, vp, vp, ctypes.c_int64, vp]
    lib.vibo_decode_mean.restype = ctypes.c_int
    lib.vibo_decode_mean.argtypes = [dp, ctypes.c_int, fp, fp, fp, vp]
    lib.vibo_decoder_person_chunks.restype = ctypes.c_int
    lib.vibo_decoder_person_chunks.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.vibo_decoder_fwd_bwd.restype = ctypes.c_int
    lib.vibo_decoder_fwd_bwd.argtypes = [ctypes.POINTER(ViboDecoderDesc)] + [vp] * 19 + [vp]
    lib.vibo_flow_stack_forward.restype = ctypes.c_int
    lib.vibo_flow_stack_forward.argtypes = [ctypes.c_int] * 3 + [vp] * 5 + [vp]
    lib.vibo_flow_stack_backward.restype = ctypes.c_int
    lib.vibo_flow_stack_backward.argtypes = [ctypes.c_int] * 3 + [vp] * 7 + [vp]
    lib.vibo_ctrain_param_floats.restype = ctypes.c_int64
    lib.vibo_ctrain_param_floats.argtypes = [dp, ctypes.c_int]
    lib.vibo_ctrain_scratch_floats.restype = ctypes.c_int64
    lib.vibo_ctrain_scratch_floats.argtypes = [dp, ctypes.c_int]
    lib.vibo_ctrain_prologue.restype = ctypes.c_int
    lib.vibo_ctrain_prologue.argtypes = [dp, ctypes.c_int, fp, fp, fp, fp, ctypes.c_uint64, ctypes.c_int, fp, ctypes.c_uint32,
                                         fp, fp, fp, fp, fp, vp, vp]
    lib.vibo_ctrain_epilogue.restype = ctypes.c_int
    lib.vibo_ctrain_epilogue.argtypes = [dp, ctypes.c_int] + [fp] * 6 + [vp] + [fp] * 9 + [vp]
    lib.vibo_code_table_scratch_bytes.restype = ctypes.c_size_t
    lib.vibo_code_table_scratch_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    for fn in (lib.vibo_code_table_sum_forward, lib.vibo_code_table_sum_backward):
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int64, fp, fp, vp, ctypes.c_size_t, vp]
    lib.vibo_train_step_supported.restype = ctypes.c_int
    lib.vibo_train_step_supported.argtypes = [dp]
    lib.vibo_elbo_fwd_bwd_step.restype = ctypes.c_int
    lib.vibo_elbo_fwd_bwd_step.argtypes = [dp, vp, ctypes.c_int, fp, vp, i64p, fp, fp, fp,              # step counter, skip, inputs
                                           fp, fp, fp, fp, fp, fp,                                   # scalars, posterior, grads
                                           vp, ctypes.c_size_t, vp]                                  # workspace, stream
    lib.vibo_train_epilogue_fused.restype = ctypes.c_int
    lib.vibo_train_epilogue_fused.argtypes = ([dp, ctypes.c_int, vp] + [fp] * 6 + [vp] + [fp] * 8 +
                                              [ctypes.c_uint64, fp, fp, fp, ctypes.c_int64, ctypes.c_uint32, vp])
    lib.vibo_train_prime.restype = ctypes.c_int
    lib.vibo_train_prime.argtypes = [dp, ctypes.c_int] + [fp] * 8 + [vp, vp]
    lib.vibo_mtrain_param_floats.restype = ctypes.c_int64
    lib.vibo_mtrain_param_floats.argtypes = [dp, ctypes.c_int]
    lib.vibo_mtrain_prologue.restype = ctypes.c_int
    lib.vibo_mtrain_prologue.argtypes = [dp, ctypes.c_int, fp, fp, fp, fp, ctypes.c_uint64, ctypes.c_int, fp, ctypes.c_uint32,
                                         fp, fp, fp, fp, vp, vp]
    lib.vibo_mean_encoder_backward_sets.restype = ctypes.c_int
    lib.vibo_mean_encoder_backward_sets.argtypes = [dp, ctypes.c_int, vp, fp, fp, fp, fp, fp, fp, ctypes.c_int, vp]
    lib.vibo_mtrain_epilogue.restype = ctypes.c_int
    lib.vibo_mtrain_epilogue.argtypes = [dp, ctypes.c_int, fp, fp, ctypes.c_int, fp, fp, fp, fp, fp, fp, vp] + [fp] * 8 + [vp]
    lib.vibo_selftest_lane_swaps.restype = ctypes.c_int
    lib.vibo_selftest_lane_swaps.argtypes = [fp, fp, vp]
    lib.vibo_set_insitu_timer.restype = ctypes.c_int
    lib.vibo_set_insitu_timer.argtypes = [vp]
    lib.vibo_insitu_timer_reset.restype = ctypes.c_int
    lib.vibo_insitu_timer_reset.argtypes = [vp, vp]
    if lib.vibo_version() != ABI_VERSION:
        raise ViboLibraryError(f'ABI mismatch: library {lib.vibo_version()} != binding {ABI_VERSION}')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().vibo_last_error_string().decode('utf-8', 'replace')
        raise ViboLibraryError(f'{what} failed (rc={rc}): {msg}')
