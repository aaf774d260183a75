"""ctypes binding of ``libsup3r_hip.so`` (include/sup3r_hip.h).

This is the only compute back-end of the package: there is no CPU fallback.
If the shared library is missing or a call fails, a ``RuntimeError`` naming the
problem is raised — nothing is silently routed elsewhere.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SUP3R_AMD_LIB: alternative build of the same library (A/B kernel probes)
LIB_PATH = os.environ.get('SUP3R_AMD_LIB') or os.path.join(
    HERE, 'lib', 'libsup3r_hip.so')

# enums (include/sup3r_hip.h)
PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2
LOSS_MAE, LOSS_MSE, LOSS_EXP = 0, 1, 2
BUF_W, BUF_G, BUF_M, BUF_V = 0, 1, 2, 3
OPT_ADAM = 0
E_TIMEOUT = -6
PRECISIONS = {'f32': PREC_F32, 'bf16': PREC_BF16, 'bf16x3': PREC_BF16X3}

EXPORTS = [
    's3_ctx_create', 's3_ctx_destroy', 's3_last_error', 's3_ctx_sync',
    's3_ctx_stream', 's3_ctx_stat', 's3_params_create', 's3_params_destroy',
    's3_params_total', 's3_params_set', 's3_params_get', 's3_params_dptr',
    's3_params_zero_grad', 's3_params_version', 's3_params_mean_abs',
    's3_adam_step', 's3_optimizer_step', 's3_optimizer_stage',
    's3_optimizer_step_staged', 's3_params_touch', 's3_capture_begin',
    's3_capture_end', 's3_capture_abort', 's3_graph_launch', 's3_graph_nodes',
    's3_graph_destroy',
    's3_ctx_set_option', 's3_ctx_get_option', 's3_option_name_at',
    's3_plan_create', 's3_plan_create_opt', 's3_plan_destroy', 's3_plan_forward',
    's3_plan_supports_window', 's3_plan_forward_window',
    's3_plan_backward', 's3_plan_tensor', 's3_plan_workspace_bytes',
    's3_plan_profile_begin', 's3_plan_profile_end',
    's3_plan_op_is_mfma', 's3_plan_op_info', 's3_plan_tensor_dtype', 's3_plan_tensor_read',
    's3_loss_content', 's3_loss_content_masked', 's3_loss_rel_bce',
    's3_lossmap_fwd', 's3_lossmap_bwd', 's3_loss_mmd',
    's3_loss_sliced_wasserstein', 's3_sw_directions', 's3_time_window',
    's3_time_mean', 's3_dft_axis',
    's3_specmap',
    's3_copy_channels', 's3_affine_channels', 's3_fill', 's3_copy_block',
    's3_coarsen', 's3_gaussian_smooth', 's3_chunk_stats',
    's3_chunk_epilogue', 's3_chunk_time_last', 's3_chunk_time_first',
    's3_step_handover', 's3_broadcast_axis',
    's3_host_register', 's3_host_unregister', 's3_d2h_window',
    's3_d2h_stream', 's3_host_alloc', 's3_host_free', 's3_d2h_async',
    's3_dma_d2h_begin', 's3_dma_wait',
    's3_invert_uv', 's3_clip_channels', 's3_range_mask', 's3_fill_indexed',
    's3_comm_unique_id', 's3_comm_init', 's3_params_allreduce_grads',
    's3_params_arm_allreduce',
    's3_allreduce_sum', 's3_params_broadcast', 's3_broadcast',
    's3_comm_destroy', 's3_comm_wait', 's3_comm_info', 's3_version',
]


OPTION_UNSET = -2 ** 31


class PlanOptions(C.Structure):
    _fields_ = [('n', C.c_int32), ('names', C.POINTER(C.c_char_p)),
                ('values', C.POINTER(C.c_int32))]


class TensorDesc(C.Structure):
    _fields_ = [('dims', C.c_int64 * 5)]


class OpDesc(C.Structure):
    _fields_ = [
        ('kind', C.c_int32), ('in0', C.c_int32), ('in1', C.c_int32),
        ('res', C.c_int32), ('out', C.c_int32), ('w', C.c_int32),
        ('b', C.c_int32), ('k', C.c_int32 * 3), ('stride', C.c_int32 * 3),
        ('lo', C.c_int32 * 3), ('hi', C.c_int32 * 3), ('pad_mode', C.c_int32),
        ('act', C.c_int32), ('alpha', C.c_float), ('d2s', C.c_int32),
        ('rep', C.c_int32), ('bcast_c', C.c_int32),
        ('reserved', C.c_int32 * 4),
    ]


_lib = None


def build_hint():
    return ('build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` or `make -C sup3r_amd/csrc` (needs hipcc)')


def lib():
    """Load (once) and return the ctypes handle of libsup3r_hip.so."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'sup3r_amd: HIP library {LIB_PATH} not found; {build_hint()}. '
            'There is no CPU fallback.')
    # torch first: its wheel bundles its own libamdhip64 / libhsa-runtime64.
    # Loading ours (linked against /opt/rocm) before torch's leaves two HIP
    # runtimes in the process and hipSetDevice then reports "no ROCm-capable
    # device"; after torch, the already-loaded runtime is shared.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    u64 = C.c_uint64
    pf = C.POINTER(C.c_float)
    sig = {
        's3_ctx_create': (i32, [i32, vp, i32, C.POINTER(vp)]),
        's3_ctx_destroy': (None, [vp]),
        's3_last_error': (C.c_char_p, [vp]),
        's3_ctx_sync': (i32, [vp]),
        's3_ctx_stream': (vp, [vp]),
        's3_ctx_stat': (i64, [vp, i32]),
        's3_params_create': (i32, [vp, i32, C.POINTER(i64), C.POINTER(vp)]),
        's3_params_destroy': (None, [vp]),
        's3_params_total': (i64, [vp]),
        's3_params_set': (i32, [vp, i32, i32, pf]),
        's3_params_get': (i32, [vp, i32, i32, pf]),
        's3_params_dptr': (vp, [vp, i32, i32]),
        's3_params_zero_grad': (i32, [vp]),
        's3_params_version': (u64, [vp]),
        's3_params_mean_abs': (i32, [vp, i32, i32, pf]),
        's3_adam_step': (i32, [vp, f32, f32, f32, f32, i64]),
        's3_optimizer_step': (i32, [vp, i32, C.POINTER(C.c_double), i32, i64]),
        's3_optimizer_stage': (i32, [vp, i32, C.POINTER(C.c_double), i32, i64]),
        's3_optimizer_step_staged': (i32, [vp, i32]),
        's3_params_touch': (i32, [vp]),
        's3_capture_begin': (i32, [vp]),
        's3_capture_end': (i32, [vp, C.POINTER(vp)]),
        's3_capture_abort': (i32, [vp]),
        's3_graph_launch': (i32, [vp]),
        's3_graph_nodes': (i64, [vp]),
        's3_graph_destroy': (None, [vp]),
        's3_plan_create': (i32, [vp, vp, C.POINTER(TensorDesc), i32,
                                 C.POINTER(OpDesc), i32, C.POINTER(i32), i32,
                                 i32, i32, i32, C.POINTER(vp)]),
        's3_plan_create_opt': (i32, [vp, vp, C.POINTER(TensorDesc), i32,
                                     C.POINTER(OpDesc), i32, C.POINTER(i32),
                                     i32, i32, i32, i32,
                                     C.POINTER(PlanOptions), C.POINTER(vp)]),
        's3_ctx_set_option': (i32, [vp, C.c_char_p, i32]),
        's3_ctx_get_option': (i32, [vp, C.c_char_p, C.POINTER(i32)]),
        's3_option_name_at': (C.c_char_p, [i32]),
        's3_plan_destroy': (None, [vp]),
        's3_plan_forward': (i32, [vp, C.POINTER(vp), vp]),
        's3_plan_supports_window': (i32, [vp]),
        's3_plan_forward_window': (i32, [vp, C.POINTER(vp), vp,
                                         C.POINTER(i64), C.POINTER(i64), vp,
                                         i32]),
        's3_plan_backward': (i32, [vp, vp, vp, i32, i32]),
        's3_plan_tensor': (vp, [vp, i32]),
        's3_plan_workspace_bytes': (i64, [vp]),
        's3_plan_profile_begin': (i32, [vp, i32]),
        's3_plan_profile_end': (i32, [vp, pf, i32]),
        's3_plan_op_is_mfma': (i32, [vp, i32]),
        's3_plan_op_info': (i32, [vp, i32, C.POINTER(i32), i32]),
        's3_plan_tensor_dtype': (i32, [vp, i32]),
        's3_plan_tensor_read': (i64, [vp, i32, vp, C.c_size_t]),
        's3_loss_content': (i32, [vp, i32, vp, i32, vp, i32, i32, i64, f32,
                                  vp, vp, i32]),
        's3_loss_content_masked': (i32, [vp, i32, vp, i32, vp, i32, vp, i32,
                                         i32, i64, f32, vp, vp, i32]),
        's3_loss_rel_bce': (i32, [vp, vp, vp, i32, f32, vp, vp, vp]),
        's3_lossmap_fwd': (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32,
                                 i32, i32, i32, vp, vp]),
        's3_lossmap_bwd': (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32,
                                 i32, i32, i32, i32, vp, vp]),
        's3_loss_mmd': (i32, [vp, vp, i32, vp, i32, i32, i64, i32, f32, f32,
                              vp, vp]),
        's3_loss_sliced_wasserstein': (i32, [vp, vp, i32, vp, i32, i32, i64,
                                             i32, i32, u64, f32, vp, vp]),
        's3_sw_directions': (i32, [vp, u64, i32, i64, vp]),
        's3_time_window': (i32, [vp, vp, i64, i32, i32, i32, i32, vp, i32,
                                 f32]),
        's3_time_mean': (i32, [vp, vp, i64, i32, i32, i32, i32, vp, i32, f32]),
        's3_dft_axis': (i32, [vp, vp, vp, vp, vp, i64, i32, i64, i32]),
        's3_specmap': (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32,
                             i32, vp, vp]),
        's3_copy_channels': (i32, [vp, vp, i32, i32, vp, i32, i32, i32, i64,
                                   i32]),
        's3_affine_channels': (i32, [vp, vp, vp, i32, i64, pf, pf]),
        's3_fill': (i32, [vp, vp, i64, f32]),
        's3_copy_block': (i32, [vp, vp, vp, i64, i64, i64, i64, i64, i64, i64]),
        's3_chunk_stats': (i32, [vp, vp, i32, i64, i32, vp]),
        's3_chunk_epilogue': (i32, [vp, vp, i32, C.POINTER(i64),
                                    C.POINTER(i64), C.POINTER(i64), i32, pf,
                                    pf, vp, vp]),
        's3_chunk_time_first': (i32, [vp, vp, i32, C.POINTER(i64), i32,
                                      C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), i32, vp]),
        's3_broadcast_axis': (i32, [vp, vp, i64, i64, i64, i64, vp]),
        's3_step_handover': (i32, [vp, vp, i64, i32, C.POINTER(i32), i32, pf,
                                   pf, vp, i32, pf, pf, vp]),
        's3_chunk_time_last': (i32, [vp, vp, i32, C.POINTER(i64),
                                     C.POINTER(i64), C.POINTER(i64), i32, pf,
                                     pf, vp]),
        's3_invert_uv': (i32, [vp, vp, i64, i64, i32, i32, i32, vp, vp]),
        's3_clip_channels': (i32, [vp, vp, i32, i64, pf, pf]),
        's3_range_mask': (i32, [vp, vp, i32, i32, i64, f32, f32, vp]),
        's3_fill_indexed': (i32, [vp, vp, i32, i32, i64, vp, vp]),
        's3_host_register': (i32, [vp, vp, C.c_size_t]),
        's3_host_unregister': (i32, [vp, vp]),
        's3_d2h_window': (i32, [vp, vp, vp, i64, i64, i64, i64, i64, vp]),
        's3_d2h_stream': (i32, [vp, vp, vp, C.c_size_t, vp, i32]),
        's3_host_alloc': (i32, [vp, C.c_size_t, i32, C.POINTER(vp)]),
        's3_host_free': (i32, [vp, vp]),
        's3_d2h_async': (i32, [vp, vp, vp, C.c_size_t, vp]),
        's3_dma_d2h_begin': (i32, [vp, vp, vp, C.c_size_t,
                                   C.POINTER(C.c_uint64)]),
        's3_dma_wait': (i32, [vp, C.c_uint64, i32]),
        's3_coarsen': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                             vp]),
        's3_gaussian_smooth': (i32, [vp, vp, i32, i32, i32, i32, i32, pf, i32,
                                     C.c_uint32, vp, vp]),
        's3_comm_unique_id': (i32, [vp]),
        's3_comm_init': (i32, [vp, i32, i32, vp]),
        's3_params_allreduce_grads': (i32, [vp]),
        's3_params_arm_allreduce': (i32, [vp, i64]),
        's3_allreduce_sum': (i32, [vp, vp, i64]),
        's3_comm_info': (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        's3_params_broadcast': (i32, [vp, i32, i32]),
        's3_broadcast': (i32, [vp, vp, i64, i32]),
        's3_comm_destroy': (None, [vp]),
        's3_comm_wait': (i32, [vp, i64]),
        's3_version': (C.c_char_p, []),
    }
    for name in EXPORTS:
        fn = getattr(L, name)   # AttributeError if the .so lacks a symbol
        fn.restype, fn.argtypes = sig[name]
    _lib = L
    return L


STATS = {'persist_dgrad': 0, 'gconv_splitk': 1, 'bucket_elems': 2,
         'dgrad_c2_slide': 3, 'buckets': 4, 'allreduces': 5}


def option_names():
    """names of the context / plan options (include/sup3r_hip.h)"""
    L, out, i = lib(), [], 0
    while True:
        name = L.s3_option_name_at(i)
        if not name:
            return out
        out.append(name.decode())
        i += 1


def plan_options(options):
    """dict name -> int | None (None = unset) as an ``s3_plan_options``
    (+ the ctypes arrays it points into, to be kept alive by the caller)"""
    items = sorted((options or {}).items())
    n = len(items)
    names = (C.c_char_p * max(n, 1))(*[k.encode() for k, _ in items])
    values = (C.c_int32 * max(n, 1))(*[
        OPTION_UNSET if v is None else int(v) for _, v in items])
    return PlanOptions(n, names, values), (names, values)

# s3_plan_op_info fields / codes (include/sup3r_hip.h)
OPINFO_FIELDS = ('kind', 'fwd', 'in16', 'out16', 'res16', 'fwd_bf16_ops',
                 'wgrad', 'dgrad', 'mask_fused_from', 'in_rep', 'res_rep',
                 'dgrad_frame16', 'fewpos_mfma')
FWD_KERNELS = ('direct', 'mfma_tile', 'mfma_persist', 'gconv', 'gconv_fewch',
               'halo32', 'fewpos', 'tail_mfma', 'small', 'fused2d', 'halo_s2',
               'mfma_gen', 'conv2d_ws', 'conv2d_head')
WGRAD_KERNELS = ('direct', 'f32_trunk', 'bf16_trunk', 'f32_gen', 'bf16_gen',
                 'bf16_2d', 'c2', 'tail', 'fewpos')
DGRAD_KERNELS = ('direct', 'mfma_frame', 'mfma_valid', 'mfma_chunked',
                 'fewch_frame', 's2', 'c2', 'gconv', 'fewpos')

TC_METHODS = {'subsample': 0, 'average': 1, 'total': 2, 'max': 3, 'min': 4}


def last_error(ctx):
    """text of the context's last error ('' if none)"""
    raw = lib().s3_last_error(ctx) if ctx else None
    return raw.decode() if raw else ''


def check(rc, ctx=None, what=''):
    if rc == 0:
        return
    msg = ''
    if ctx:
        raw = lib().s3_last_error(ctx)
        msg = raw.decode() if raw else ''
    if rc == E_TIMEOUT:
        raise TimeoutError(f'sup3r_amd: {msg or what}')
    raise RuntimeError(f'sup3r_amd HIP call failed ({what}, code {rc}): {msg}')
