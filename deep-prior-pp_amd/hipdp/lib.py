"""
ctypes binding of the C ABI declared in include/dpp_hip.h.

`load()` returns the product library deep-prior-pp_amd/lib/libdpp_hip.so and raises (loudly) when it is
missing -- there is no CPU fallback in the product.  `load(path)` lets the CPU-side kernel-logic tests
hand in the emulator build of the very same sources (tests/emu/_build/libdpp_emu.so).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(_HERE), 'lib', 'libdpp_hip.so')

ABI_VERSION = 11
ST_A, ST_B, ST_C, ST_BNX = 1, 2, 4, 8      # DPP_ST_*: which pointers of a call address bf16-stored activation tensors
c_float_p = C.c_void_p      # device pointers travel as integers
stream_t = C.c_void_p


class RowMap(C.Structure):
    _fields_ = [('s', C.c_int), ('Wo', C.c_int), ('HoWo', C.c_int), ('Wi', C.c_int), ('HiWi', C.c_int)]

    @staticmethod
    def identity():
        return RowMap(1, 0, 0, 0, 0)

    @staticmethod
    def strided(s, Ho, Wo, Hi, Wi):
        return RowMap(s, Wo, Ho * Wo, Wi, Hi * Wi)


class Act(C.Structure):
    _fields_ = [('mean', C.c_void_p), ('scale', C.c_void_p), ('beta', C.c_void_p), ('mode', C.c_int), ('cmod', C.c_int),
                ('x2', C.c_void_p), ('aux', C.c_void_p), ('out', C.c_void_p)]

    NONE, RELU, BN, BN_RELU, BN_BWD = 0, 1, 2, 3, 4

    @staticmethod
    def none():
        return Act(None, None, None, 0, 1, None, None, None)


class Epilogue(C.Structure):
    _fields_ = [('stats', C.c_void_p), ('bn_x', C.c_void_p), ('bn_mean', C.c_void_p), ('bn_inv_std', C.c_void_p),
                ('bn_scale', C.c_void_p), ('bn_beta', C.c_void_p), ('bn_relu', C.c_int), ('pad_', C.c_int),
                ('bn_partial', C.c_void_p)]


class GemmDesc(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int), ('a_kc', C.c_int), ('mapA', RowMap), ('actA', Act),
                ('B', C.c_void_p), ('ldb', C.c_int), ('b_kc', C.c_int), ('mapB', RowMap), ('actB', Act),
                ('C', C.c_void_p), ('ldc', C.c_int), ('mapC', RowMap),
                ('bias', C.c_void_p), ('residual', C.c_void_p),
                ('M', C.c_int), ('N', C.c_int), ('K', C.c_int),
                ('splitk', C.c_int), ('partial', C.c_void_p),
                ('bm', C.c_int), ('bn', C.c_int), ('wm', C.c_int), ('variant', C.c_int), ('epi', Epilogue), ('store', C.c_int), ('precision', C.c_int)]


class BnEval(C.Structure):
    _fields_ = [('mean', C.c_void_p), ('inv_std', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p)]


class ResblockDesc(C.Structure):
    _fields_ = [('X', C.c_void_p), ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int),
                ('stride', C.c_int), ('Ho', C.c_int), ('Wo', C.c_int), ('Cout', C.c_int), ('Nb', C.c_int),
                ('bn0', BnEval), ('bn1', BnEval), ('bn2', BnEval),
                ('W1', C.c_void_p), ('b1', C.c_void_p), ('W2', C.c_void_p), ('b2', C.c_void_p), ('W3', C.c_void_p), ('b3', C.c_void_p),
                ('Wsc', C.c_void_p), ('bsc', C.c_void_p), ('Y', C.c_void_p), ('store', C.c_int)]


# name -> (restype, argtypes); every symbol include/dpp_hip.h declares must be listed here
SIGNATURES = {
    'dpp_abi_version': (C.c_int, []),
    'dpp_gemm': (C.c_int, [C.POINTER(GemmDesc), stream_t]),
    'dpp_gemm_variant_rows': (C.c_int, [C.POINTER(GemmDesc)]),
    'dpp_fc_gemm': (C.c_int, [C.POINTER(GemmDesc), C.c_int, C.c_int, stream_t]),
    'dpp_reduce_partials': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, stream_t]),
    'dpp_conv3x3': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Act), C.c_void_p, C.c_int,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Epilogue), C.c_int, stream_t]),
    'dpp_conv3x3_bf16': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Act), C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Epilogue), C.c_int, stream_t]),
    'dpp_conv3x3_tiling': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'dpp_conv3x3_stream_rows': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'dpp_conv3x3_stream': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Act), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(Epilogue), stream_t]),
    'dpp_conv3x3_wtrans': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, stream_t]),
    'dpp_conv3x3_wgrad_blocks': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'dpp_conv3x3_wgrad': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Act), C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int, C.c_int, stream_t]),
    'dpp_conv3x3_wgrad_bf16': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Act), C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int, C.c_int, stream_t]),
    'dpp_conv3x3_wgrad_bf16_ok': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'dpp_stem_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_int, stream_t]),
    'dpp_stem_wgrad_blocks': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'dpp_stem_wgrad': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_int, stream_t]),
    'dpp_convpool_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, stream_t]),
    'dpp_convpool_wgrad_blocks': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'dpp_convpool_wgrad': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, stream_t]),
    'dpp_convpool_dgrad': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p, stream_t]),
    'dpp_bn_stats_partial': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, stream_t]),
    'dpp_bn_finalize': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, stream_t]),
    'dpp_bn_eval_coeffs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     stream_t]),
    'dpp_bn_eval_job_bytes': (C.c_size_t, []),
    'dpp_bn_eval_coeffs_multi': (C.c_int, [C.c_void_p, C.c_int, C.c_int, stream_t]),
    'dpp_resblock_eval_ok': (C.c_int, [C.c_int] * 5),
    'dpp_resblock_eval': (C.c_int, [C.POINTER(ResblockDesc), stream_t]),
    'dpp_resblock_eval_check': (C.c_int, [C.POINTER(ResblockDesc)]),
    'dpp_bn_bwd_reduce': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, stream_t]),
    'dpp_bn_bwd_finalize': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, stream_t]),
    'dpp_bn_bwd_finalize_apply_ok': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'dpp_bn_bwd_finalize_apply': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, stream_t]),
    'dpp_bn_bwd_apply': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, stream_t]),
    'dpp_reduce_job_bytes': (C.c_size_t, []),
    'dpp_reduce_multi_block_cols': (C.c_int, []),
    'dpp_reduce_multi': (C.c_int, [C.c_void_p, C.c_int, C.c_int, stream_t]),
    'dpp_colsum_partial': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, stream_t]),
    'dpp_wgrad_stream_slices': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'dpp_wgrad_stream': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, stream_t]),
    'dpp_fc_wgrad_stream_ok': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'dpp_fc_wgrad_stream': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, stream_t]),
    'dpp_wgrad_stream_bf16': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, stream_t]),
    'dpp_wgrad_stream_bf16_ok': (C.c_int, [C.c_int, C.c_int]),
    'dpp_wgrad3_stream_slices': (C.c_int, [C.c_int] * 6),
    'dpp_wgrad3_stream': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, stream_t]),
    'dpp_wtrans_job_bytes': (C.c_size_t, []),
    'dpp_conv3x3_wtrans_multi': (C.c_int, [C.c_void_p, C.c_int, C.c_int, stream_t]),
    'dpp_loss_sse': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, stream_t]),
    'dpp_reduce_partials_loss': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, stream_t]),
    'dpp_loss_sse_bcast': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, stream_t]),
    'dpp_error_l2': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, stream_t]),
    'dpp_adam': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, stream_t]),
    'dpp_adam_ticked': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, stream_t]),
    'dpp_adam_tick': (C.c_int, [C.c_void_p, stream_t]),
    'dpp_counter_add': (C.c_int, [C.c_void_p, C.c_ulonglong, stream_t]),
    'dpp_axpy': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_size_t, stream_t]),
    'dpp_sumsq': (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_int, stream_t]),
    'dpp_sumsq_multi_workspace_bytes': (C.c_size_t, []),
    'dpp_sumsq_multi': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, stream_t]),
    'dpp_axpy_multi': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, stream_t]),
    'dpp_scale': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_size_t, stream_t]),
    'dpp_relu_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, stream_t]),
    'dpp_fill_zero': (C.c_int, [C.c_void_p, C.c_size_t, stream_t]),
    'dpp_copy2d': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, stream_t]),
    'dpp_rowscale': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, stream_t]),
    'dpp_crop_center': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, stream_t]),
    'dpp_bernoulli_mask': (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_ulonglong, C.c_ulonglong, C.c_void_p, stream_t]),
    'dpp_augment_record_bytes': (C.c_size_t, []),
    'dpp_augment_prepare': (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p] * 5 + [C.c_int, C.c_ulonglong,
                                      C.c_ulonglong, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, stream_t]),
    'dpp_augment_warp': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, stream_t]),
    'dpp_augment': (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p] * 5 + [C.c_int, C.c_ulonglong,
                              C.c_ulonglong, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                              C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int,
                              stream_t]),
    'dpp_crop_record_bytes': (C.c_size_t, []),
    'dpp_crop_prepare': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, stream_t]),
    'dpp_crop_com_workspace_bytes': (C.c_size_t, [C.c_int]),
    'dpp_crop_com': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, stream_t]),
    'dpp_crop_warp': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, stream_t]),
    'dpp_crop_refine': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, stream_t]),
    'dpp_pose_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_long, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, stream_t]),
    'dpp_pose_sample_rot3d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_long, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, stream_t]),
    'dpp_pca_workspace_bytes': (C.c_size_t, [C.c_long, C.c_int]),
    'dpp_pca_fit': (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, stream_t]),
    'dpp_pose_eval': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, stream_t]),
    'dpp_prof_set': (C.c_int, [C.c_void_p]),
    'dpp_plan_create': (C.c_int, [C.POINTER(C.c_void_p)]),
    'dpp_plan_destroy': (C.c_int, [C.c_void_p]),
    'dpp_plan_record_begin': (C.c_int, [C.c_void_p]),
    'dpp_plan_record_lane': (C.c_int, [C.c_void_p, C.c_int]),
    'dpp_plan_record_end': (C.c_int, [C.c_void_p]),
    'dpp_plan_fork': (C.c_int, [C.c_void_p]),
    'dpp_plan_join': (C.c_int, [C.c_void_p]),
    'dpp_plan_count': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'dpp_plan_run': (C.c_int, [C.c_void_p, stream_t, stream_t]),
    'dpp_plan_graph_build': (C.c_int, [C.c_void_p, C.c_int]),
    'dpp_plan_graph_launch': (C.c_int, [C.c_void_p, stream_t]),
}


class DppError(RuntimeError):
    pass


def check(status, what):
    if status != 0:
        raise DppError("%s failed with status %d" % (what, status))


_cache = {}


def load(path=None):
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise DppError("HIP kernel library not found: %s (build it with __graft_entry__.build() or "
                       "`make -C deep-prior-pp_amd/csrc hip`); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.dpp_abi_version()
    if v != ABI_VERSION:
        raise DppError("ABI version mismatch: library %d, binding %d" % (v, ABI_VERSION))
    _cache[path] = lib
    return lib
