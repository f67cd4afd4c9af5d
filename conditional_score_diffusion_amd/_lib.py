"""ctypes binding of libcsd_hip.so (the C ABI declared in include/csd.h).

PyTorch-ROCm tensors are used purely as device-memory containers: every call hands raw
``data_ptr()`` addresses and the current HIP stream to the library.  There is NO fallback: if the
shared library is missing or a call fails, a ``RuntimeError`` is raised (the
reference's pybind ops raise RuntimeError via TORCH_CHECK, op/upfirdn2d.cpp:8-10).
"""
import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CSD_LIB_PATH', os.path.join(_HERE, 'libcsd_hip.so'))   # override: tuning builds only
CSRC = os.path.join(_HERE, 'csrc')

MAX_LEVELS = 8
MAX_ATTN = 8

ACT_IDS = {'none': 0, 'swish': 1, 'relu': 2, 'lrelu': 3, 'elu': 4}
PREC_IDS = {'fp32': 0, 'f32': 0, 'fp16x3': 1, 'fp16': 2, 'fp16f8': 3}


class UNetConfig(ctypes.Structure):
    _fields_ = [('arch', ctypes.c_int32), ('nf', ctypes.c_int32), ('n_levels', ctypes.c_int32),
                ('ch_mult', ctypes.c_int32 * MAX_LEVELS), ('num_res_blocks', ctypes.c_int32),
                ('n_attn', ctypes.c_int32), ('attn_resolutions', ctypes.c_int32 * MAX_ATTN),
                ('image_size', ctypes.c_int32), ('x_channels', ctypes.c_int32),
                ('y_channels', ctypes.c_int32), ('out_channels', ctypes.c_int32),
                ('resamp_with_conv', ctypes.c_int32), ('conditional', ctypes.c_int32),
                ('centered', ctypes.c_int32), ('act', ctypes.c_int32), ('precision', ctypes.c_int32),
                ('skip_rescale', ctypes.c_int32), ('progressive', ctypes.c_int32), ('progressive_input', ctypes.c_int32),
                ('embedding_type', ctypes.c_int32), ('n_fir', ctypes.c_int32), ('fir_kernel', ctypes.c_float * 8)]


class PCParams(ctypes.Structure):
    _fields_ = [('n_steps', ctypes.c_int32), ('labels', ctypes.POINTER(ctypes.c_float)),
                ('std_x', ctypes.POINTER(ctypes.c_float)), ('G', ctypes.POINTER(ctypes.c_float)),
                ('std_y', ctypes.POINTER(ctypes.c_float)), ('snr', ctypes.c_float),
                ('denoise', ctypes.c_int32), ('noise_tape', ctypes.c_void_p), ('seed', ctypes.c_uint64),
                ('record', ctypes.c_void_p), ('predictor', ctypes.c_int32), ('corrector', ctypes.c_int32),
                ('pred_coef', ctypes.POINTER(ctypes.c_float)), ('corr_coef', ctypes.POINTER(ctypes.c_float)),
                ('path_coef', ctypes.POINTER(ctypes.c_float)), ('path_std0', ctypes.c_float),
                ('corr_alpha', ctypes.POINTER(ctypes.c_float))]


def build(verbose=False):
    """Compile libcsd_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ['make', '-C', CSRC, '-j', str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError('building libcsd_hip.so failed (see output above)')
    return LIB_PATH


_lib = None
WEIGHT_EPOCH = [0]     # bumped by every raw-kernel write to model parameters (optimizer step, EMA copy): invalidates packed weights

# name -> (restype, argtypes); mirrors include/csd.h one to one
_vp, _i, _i64, _f, _sz, _u64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                ctypes.c_size_t, ctypes.c_uint64)
SIGNATURES = {
    'csd_version': (ctypes.c_char_p, []),
    'csd_last_error': (ctypes.c_char_p, []),
    'csd_profile_start': (_i, []),
    'csd_profile_select': (_i, [ctypes.c_uint, _i]),
    'csd_profile_stop': (_i, [_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64),
                              ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'csd_profile_stop_ex': (_i, [_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'csd_unet_create': (_i, [ctypes.POINTER(UNetConfig), ctypes.POINTER(_vp)]),
    'csd_unet_destroy': (None, [_vp]),
    'csd_unet_num_params': (_i, [_vp]),
    'csd_unet_param_info': (_i, [_vp, _i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_i),
                                 ctypes.POINTER(_i64)]),
    'csd_unet_set_param': (_i, [_vp, ctypes.c_char_p, _vp, _i64]),
    'csd_unet_packed_bytes': (_sz, [_vp]),
    'csd_unet_pack': (_i, [_vp, _vp, _vp]),
    'csd_unet_workspace_bytes': (_sz, [_vp, _i]),
    'csd_unet_forward': (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp]),
    'csd_unet_stats': (_i, [_vp, _i, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double),
                            ctypes.POINTER(ctypes.c_double)]),
    'csd_pc_scratch_bytes': (_sz, [_vp, _i]),
    'csd_pc_sample': (_i, [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _i, ctypes.POINTER(PCParams), _vp]),
    'csd_pc_step_begin': (_i, [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _i, ctypes.POINTER(PCParams), _i, _vp, _vp]),
    'csd_pc_step_end': (_i, [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _i, ctypes.POINTER(PCParams), _i, _vp, _i, _vp]),
    'csd_unet_train_workspace_bytes': (_sz, [_vp, _i, _f]),
    'csd_unet_train_forward': (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _i, _f, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    'csd_unet_backward': (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _i, ctypes.c_uint64, _vp]),
    'csd_unet_train_release': (_i, [_vp, _vp]),
    'csd_unet_train_release_call': (_i, [_vp, _vp, ctypes.c_uint64]),
    'csd_unet_backward_marks': (_i, [_vp, _vp, _vp, _i]),
    'csd_unet_backward_marks_epoch': (ctypes.c_uint64, [_vp]),
    'csd_event_create': (_vp, []),
    'csd_event_destroy': (_i, [_vp]),
    'csd_stream_wait_event': (_i, [_vp, _vp]),
    'csd_event_query': (_i, [_vp]),
    'csd_update_scratch_bytes': (_sz, [_i]),
    'csd_langevin_step': (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _i, _i64, _vp, _vp]),
    'csd_reverse_diffusion_step': (_i, [_vp, _vp, _vp, _vp, _f, _f, _i, _i64, _vp]),
    'csd_row_norms': (_i, [_vp, _vp, _i, _i64, _vp]),
    'csd_affine_noise_step': (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _i64, _vp]),
    'csd_linear': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'csd_fourier_embedding': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'csd_axpby': (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _i64, _vp]),
    'csd_bias_add_nchw': (_i, [_vp, _vp, _vp, _i, _i, _i64, _i, _i, _vp]),
    'csd_randn': (_i, [_vp, _i64, _u64, _u64, _vp]),
    'csd_scale_rows': (_i, [_vp, _vp, _vp, _i, _i, _i64, _vp]),
    'csd_groupnorm_scratch_bytes': (_sz, [_i, _i, _i, _i]),
    'csd_groupnorm_act': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    'csd_conv_scratch_bytes': (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    'csd_conv2d': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'csd_attention_scratch_bytes': (_sz, [_i, _i, _i, _i]),
    'csd_attention': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'csd_upfirdn2d': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'csd_fused_bias_act': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i64, _i, _i, _f, _f, _vp]),
    'csd_nearest_up2': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'csd_timestep_embedding': (_i, [_vp, _vp, _i, _i, _vp]),
    'csd_conv_wgrad_scratch_bytes': (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    'csd_conv2d_wgrad': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'csd_groupnorm_act_backward': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    'csd_attention_backward_scratch_bytes': (_sz, [_i, _i, _i, _i]),
    'csd_attention_backward': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'csd_bgemm': (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i64, _i64, _i64, _f, _vp]),
    'csd_sum_inner': (_i, [_vp, _vp, _i64, _i64, _vp]),
    'csd_sum_rows': (_i, [_vp, _vp, _i, _i, _vp]),
    'csd_act': (_i, [_vp, _vp, _vp, _i, _i64, _vp]),
    'csd_mul': (_i, [_vp, _vp, _vp, _i64, _vp]),
    'csd_conv3x3_block_scratch_bytes': (_sz, [_i, _i]),
    'csd_conv3x3_block': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'csd_conv2d_ex': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'csd_conv2d_wgrad_ex': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'csd_groupnorm_nhwc_scratch_bytes': (_sz, [_i, _i, _i]),
    'csd_groupnorm_act_nhwc': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    'csd_groupnorm_act_backward_nhwc': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'csd_bias_add_nhwc': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'csd_sum_pixels_scratch_bytes': (_sz, [_i, _i, _i]),
    'csd_sum_pixels_nhwc': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'csd_zero_insert_odd_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'csd_sumpool2_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'csd_attention_nhwc': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'csd_attention_nhwc_prec': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'csd_attention_backward_nhwc': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'csd_adam_step': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _f, _f, _f, _f, _f, _f, _vp]),
    'csd_global_norm_scratch_bytes': (_sz, []),
    'csd_global_norm': (_i, [_vp, _vp, _i64, _vp, _vp]),
    'csd_ema_update': (_i, [_vp, _vp, _i64, _f, _vp]),
    'csd_dropout': (_i, [_vp, _vp, _vp, _f, _u64, _u64, _i64, _vp]),
}


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libcsd_hip.so not found at %s - run `python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C %s`). There is no CPU fallback.' % (LIB_PATH, CSRC))
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)   # AttributeError here = header/library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


PROF_CLASSES = ['conv3x3', 'conv3x3_resample', 'conv1x1', 'gn_stats', 'gn_finalize', 'attention', 'sampler', 'other', 'gn_apply16',
                'conv3x3_other']


def profile_start():
    check(lib().csd_profile_start(), 'profile_start')


def profile_select(classes=None, step_stride=1):
    """Limit the profiler's events to the named launch classes (None = all) on every step_stride-th PC step."""
    mask = 0xFFFFFFFF if classes is None else sum(1 << PROF_CLASSES.index(c) for c in classes)
    check(lib().csd_profile_select(mask, int(step_stride)), 'profile_select')


def profile_stop():
    """-> {class: {'ms', 'launches', 'flops', 'bytes', 'alg_bytes'}} for the launches since profile_start(): `bytes` = what the kernels
    have to move (operand planes, residual reads), `alg_bytes` = SURVEY.md 8(d): input + output tensor of the layer, fp32."""
    n = len(PROF_CLASSES)
    ms, fl, by, ab = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
    la = (ctypes.c_int64 * n)()
    check(lib().csd_profile_stop_ex(n, ms, la, fl, by, ab), 'profile_stop')
    return {PROF_CLASSES[i]: {'ms': ms[i], 'launches': la[i], 'flops': fl[i], 'bytes': by[i], 'alg_bytes': ab[i]} for i in range(n)}


ERR_NONFINITE = -6            # include/csd.h CSD_ERR_NONFINITE


class NonFiniteError(FloatingPointError, RuntimeError):
    """the fused sampler's state (or a Langevin norm) left the finite range: an fp16-operand arithmetic mode met an operand beyond
    65504 - run the network with ``config.model.csd_precision = 'fp32'`` (include/csd.h, csd_pc_sample's finiteness contract)"""


def check(rc, what=''):
    if rc != 0:
        msg = lib().csd_last_error().decode('utf-8', 'replace')
        if rc == ERR_NONFINITE:
            raise NonFiniteError('libcsd_hip %s: %s' % (what, msg))
        raise RuntimeError('libcsd_hip %s failed (status %d): %s' % (what, rc, msg))


def ptr(t):
    """Device address of a torch tensor (None -> NULL). The tensor must be contiguous fp32 on a GPU."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise RuntimeError('libcsd_hip needs contiguous tensors')
    return ctypes.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu_tensor(t, name='tensor'):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError('%s must be a float32 tensor on the MI355X (got %s on %s); the HIP path has no CPU '
                           'fallback' % (name, getattr(t, 'dtype', type(t)), getattr(t, 'device', '?')))
