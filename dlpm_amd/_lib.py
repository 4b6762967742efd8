"""ctypes binding of libdlpm_amd.so (include/dlpm_amd.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Device pointers are plain integers (`tensor.data_ptr()`); streams are `hipStream_t` values
(`torch.cuda.current_stream().cuda_stream`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DLPM_LIB') or os.path.join(_HERE, 'lib', 'libdlpm_amd.so')   # DLPM_LIB: an instrumented build (dlpm_amd/build.py)

vp, i32, i64, u32, u64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double

ABI_VERSION = 6   # must equal dlpm_abi_version(): struct layouts below mirror include/dlpm_amd.h at this version
UPD_DLIM, UPD_CLIP, UPD_ADVANCE, SMP_NO_FUSED_MLP, UPD_ELEMENTWISE, SMP_LIM = 1, 2, 4, 8, 16, 32
MEAN_TYPES = {'EPSILON': 0, 'START_X': 1, 'Z': 2, 'PREVIOUS_X': 3}   # dlpm_mean_type
PRED_TO_XSTART, PRED_CLIP, PRED_TO_EPS, PRED_ELEMENTWISE = 1, 2, 4, 16
CONV_AUTO, CONV_F4, CONV_F2, CONV_IGEMM = 0, 1, 2, 3
GEMM_AUTO, GEMM_F32, GEMM_BF16X3 = 0, 1, 2


class MT19937(C.Structure):
    _fields_ = [('key', u32 * 624), ('pos', i32), ('has_cached', i32), ('cached', f64)]


class UpdateArgs(C.Structure):
    _fields_ = [('x_dev', vp), ('eps_dev', vp), ('z_dev', vp), ('t_dev', vp), ('g_dev', vp), ('bg_dev', vp),
                ('bs_dev', vp), ('c_eps_dev', vp), ('c_noise_dev', vp), ('A_dev', vp), ('B', i64), ('D', i64),
                ('T', i32), ('flags', i32), ('dlim_eta', f32), ('alpha', f32), ('seed', u64), ('sample_offset', i64), ('key_dev', vp),
                ('hist_pp', vp)]


class PredictArgs(C.Structure):
    _fields_ = [('x_dev', vp), ('in_dev', vp), ('out_dev', vp), ('t_dev', vp), ('g_dev', vp), ('bg_dev', vp), ('bs_dev', vp),
                ('c_eps_dev', vp), ('A_dev', vp), ('B', i64), ('D', i64), ('T', i32), ('mean_type', i32), ('flags', i32)]


class LimUpdateArgs(C.Structure):
    _fields_ = [('x_dev', vp), ('eps_dev', vp), ('z_dev', vp), ('t_dev', vp), ('tmp_dev', vp), ('cx_dev', vp),
                ('cs_dev', vp), ('cn_dev', vp), ('A_dev', vp), ('B', i64), ('D', i64), ('T', i32), ('flags', i32),
                ('clamp_eps', f32), ('seed', u64), ('sample_offset', i64), ('key_dev', vp), ('hist_pp', vp)]


class UNetConfig(C.Structure):
    _fields_ = [('in_channels', i32), ('model_channels', i32), ('out_channels', i32), ('num_res_blocks', i32),
                ('num_heads', i32), ('image_size', i32), ('n_mult', i32), ('channel_mult', i32 * 8), ('n_attn', i32),
                ('attention_resolutions', i32 * 8)]


class ConvArgs(C.Structure):
    _fields_ = [('src0', vp), ('src1', vp), ('C0', i32), ('C1', i32), ('B', i32), ('Hin', i32), ('Win', i32),
                ('Hout', i32), ('Wout', i32), ('ksize', i32), ('stride', i32), ('upsample', i32), ('weight', vp),
                ('bias', vp), ('coefA', vp), ('coefB', vp), ('act_silu', i32), ('res0', vp), ('res1', vp), ('R0', i32),
                ('out', vp), ('Cout', i32), ('in_nchw', i32), ('out_nchw', i32), ('force_direct', i32), ('scratch_floats', i64)]


class ResBlockArgs(C.Structure):
    _fields_ = [('x0', vp), ('x1', vp), ('C0', i32), ('C1', i32), ('B', i32), ('H', i32), ('W', i32), ('gn1_w', vp), ('gn1_b', vp),
                ('conv1_w', vp), ('conv1_b', vp), ('ss', vp), ('ss_stride', i64), ('gn2_w', vp), ('gn2_b', vp), ('conv2_w', vp),
                ('conv2_b', vp), ('skip_w', vp), ('skip_b', vp), ('out', vp), ('stats_out', vp)]


class AttnBlockArgs(C.Structure):
    _fields_ = [('x', vp), ('C', i32), ('heads', i32), ('B', i32), ('H', i32), ('W', i32), ('gn_w', vp), ('gn_b', vp),
                ('qkv_w', vp), ('qkv_b', vp), ('proj_w', vp), ('proj_b', vp), ('out', vp), ('stats_out', vp)]


class SamplerConfig(C.Structure):
    _fields_ = [('unet', vp), ('mlp', vp), ('B', i64), ('C', i32), ('H', i32), ('W', i32), ('T', i32), ('alpha', f64),
                ('clamp_a', f64), ('clamp_eps', f64), ('flags', i32), ('dlim_eta', f32), ('seed', u64),
                ('sample_offset', i64), ('use_graph', i32), ('g', vp), ('bg', vp), ('s', vp), ('bs', vp),
                ('lim_ts', vp), ('lim_tmp', vp), ('lim_cx', vp), ('lim_cs', vp), ('lim_cn', vp), ('in_scale', vp),
                ('mean_type', i32)]


# name -> (restype, argtypes); one entry per function declared in include/dlpm_amd.h
SIGNATURES = {
    'dlpm_last_error': (C.c_char_p, []),
    'dlpm_abi_version': (C.c_int, []),
    'dlpm_prof_enable': (C.c_int, [C.c_int]),
    'dlpm_prof_report': (C.c_int, [C.c_char_p, i64]),
    'dlpm_schedule_f32': (C.c_int, [C.c_int, f64, vp, vp, vp, vp]),
    'dlpm_schedule_exploding_f32': (C.c_int, [C.c_int, f64, vp, vp, vp, vp]),
    'dlpm_mt19937_seed': (C.c_int, [C.POINTER(MT19937), u32]),
    'dlpm_skewed_levy_host_f32': (C.c_int, [C.POINTER(MT19937), f64, i64, f64, vp]),
    'dlpm_randn_host_f32': (C.c_int, [C.POINTER(MT19937), i64, vp]),
    'dlpm_skewed_levy_philox_f32': (C.c_int, [vp, C.c_int, i64, f64, f64, u64, i64, vp]),
    'dlpm_skewed_levy_elem_philox_f32': (C.c_int, [vp, C.c_int, i64, i64, f64, f64, u64, i64, vp]),
    'dlpm_init_state_elem_philox_f32': (C.c_int, [vp, i64, i64, f64, f64, f32, u64, i64, vp]),
    'dlpm_init_state_philox_f32': (C.c_int, [vp, i64, i64, f64, f64, f32, u64, i64, vp]),
    'dlpm_coeff_tables_f32': (C.c_int, [vp, vp, vp, vp, C.c_int, i64, vp, vp, vp, vp]),
    'dlpm_update_f32': (C.c_int, [C.POINTER(UpdateArgs), vp]),
    'dlpm_predict_f32': (C.c_int, [C.POINTER(PredictArgs), vp]),
    'dlpm_fill_scaled_t_f32': (C.c_int, [vp, vp, i32, i64, vp]),
    'dlpm_postprocess_f32': (C.c_int, [vp, vp, i64, f32, C.c_int, vp]),
    'dlpm_scale_by_table_f32': (C.c_int, [vp, vp, i64, vp, vp, vp]),
    'dlpm_lim_tables_f32': (C.c_int, [f64, i32, i32, vp, vp, vp, vp, vp]),
    'dlpm_lim_update_f32': (C.c_int, [C.POINTER(LimUpdateArgs), vp]),
    'dlpm_fill_table_t_f32': (C.c_int, [vp, vp, vp, i32, i64, vp]),
    'dlpm_images_to_rgb8': (C.c_int, [vp, vp, i64, i32, i32, i32, vp]),
    'dlpm_png_bound': (i64, [i32, i32]),
    'dlpm_png_encode_rgb8': (C.c_int, [vp, i32, i32, i32, vp, i64, C.POINTER(i64)]),
    'dlpm_png_write_rgb8': (C.c_int, [vp, i64, i32, i32, C.c_char_p, i64, i32, i32]),
    'dlpm_unet_create': (C.c_int, [C.POINTER(UNetConfig), C.POINTER(vp)]),
    'dlpm_unet_set_param': (C.c_int, [vp, C.c_char_p, vp, i64]),
    'dlpm_unet_num_params': (C.c_int, [vp]),
    'dlpm_unet_param_key': (C.c_char_p, [vp, C.c_int, C.POINTER(i64)]),
    'dlpm_unet_finalize': (C.c_int, [vp]),
    'dlpm_unet_workspace_bytes': (i64, [vp, i64]),
    'dlpm_unet_forward': (C.c_int, [vp, vp, vp, vp, i64, vp, i64, vp]),
    'dlpm_unet_forward_uniform_t': (C.c_int, [vp, vp, vp, vp, i64, vp, i64, vp]),
    'dlpm_unet_forward_update': (C.c_int, [vp, vp, vp, C.POINTER(UpdateArgs), vp, i64, vp, i64, vp]),
    'dlpm_unet_time_embedding_width': (i64, [vp]),
    'dlpm_unet_time_embeddings_scratch_bytes': (i64, [vp, i64]),
    'dlpm_unet_time_embeddings': (C.c_int, [vp, vp, i64, vp, vp, i64, vp]),
    'dlpm_unet_bind_time_table': (C.c_int, [vp, vp, vp]),
    'dlpm_unet_keep_features': (C.c_int, [vp, C.c_int]),
    'dlpm_unet_num_features': (C.c_int, [vp]),
    'dlpm_unet_feature_shape': (C.c_int, [vp, C.c_int, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    'dlpm_unet_get_feature': (C.c_int, [vp, C.c_int, vp, i64, vp]),
    'dlpm_unet_flops_per_sample': (i64, [vp]),
    'dlpm_unet_set_conv_policy': (C.c_int, [vp, i32, i64]),
    'dlpm_unet_set_gemm_policy': (C.c_int, [vp, i32]),
    'dlpm_unet_plan_version': (i64, [vp]),
    'dlpm_unet_destroy': (None, [vp]),
    'dlpm_mlp_create': (C.c_int, [i32, i32, i32, i32, C.POINTER(vp)]),
    'dlpm_mlp_set_param': (C.c_int, [vp, C.c_char_p, vp, i64]),
    'dlpm_mlp_finalize': (C.c_int, [vp]),
    'dlpm_mlp_forward': (C.c_int, [vp, vp, vp, vp, i64, vp]),
    'dlpm_mlp_sample_steps_f32': (C.c_int, [vp, vp, vp, vp, vp, i32, i64, i32, i32, u64, i64, vp, vp]),
    'dlpm_mlp_destroy': (None, [vp]),
    'dlpm_conv2d_f32': (C.c_int, [C.POINTER(ConvArgs), vp, vp]),
    'dlpm_groupnorm_coeffs_f32': (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, i64, i64, vp, vp, vp]),
    'dlpm_attention_f32': (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    'dlpm_resblock_small_f32': (C.c_int, [C.POINTER(ResBlockArgs), vp, i64, vp]),
    'dlpm_resblock_img_f32': (C.c_int, [C.POINTER(ResBlockArgs), vp, i64, vp]),
    'dlpm_resblock_img_scratch_floats': (i64, [i64, i32]),
    'dlpm_attnblock_small_f32': (C.c_int, [C.POINTER(AttnBlockArgs), vp, i64, vp]),
    'dlpm_timestep_embedding_f32': (C.c_int, [vp, vp, i64, i32, vp]),
    'dlpm_nchw_to_nhwc_f32': (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    'dlpm_nhwc_to_nchw_f32': (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    'dlpm_sampler_create': (C.c_int, [C.POINTER(SamplerConfig), C.POINTER(vp)]),
    'dlpm_sampler_reseed': (C.c_int, [vp, u64, i64]),
    'dlpm_sampler_begin': (C.c_int, [vp, vp]),
    'dlpm_sampler_begin_injected': (C.c_int, [vp, vp, vp, vp]),
    'dlpm_sampler_set_state': (C.c_int, [vp, vp, i32, vp]),
    'dlpm_sampler_step_injected': (C.c_int, [vp, vp, vp]),
    'dlpm_sampler_steps': (C.c_int, [vp, i32, vp]),
    'dlpm_sampler_set_history': (C.c_int, [vp, vp, vp]),
    'dlpm_sampler_copy_state': (C.c_int, [vp, vp, vp]),
    'dlpm_sampler_state': (vp, [vp]),
    'dlpm_sampler_t': (i32, [vp]),
    'dlpm_sampler_table': (vp, [vp, C.c_int]),
    'dlpm_sampler_destroy': (None, [vp]),
}

_lib = None


class DlpmError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DlpmError('libdlpm_amd.so is missing at %s -- run `python -m dlpm_amd.build` '
                            '(or __graft_entry__.build()); there is no CPU fallback' % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        got = handle.dlpm_abi_version()
        if got != ABI_VERSION:
            raise DlpmError('libdlpm_amd.so has ABI version %d, the Python mirror expects %d -- rebuild with '
                            '`python -m dlpm_amd.build`' % (got, ABI_VERSION))
        _lib = handle
    return _lib


def check(rc):
    """Raise with the library's message on a non-zero status (the reference raises Python exceptions)."""
    if rc != 0:
        msg = lib().dlpm_last_error().decode('utf-8', 'replace')
        if rc == -1:
            raise ValueError(msg)
        raise DlpmError('[dlpm status %d] %s' % (rc, msg))


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """data_ptr of a contiguous fp32 tensor (or None)."""
    if t is None:
        return None
    import torch
    assert t.is_contiguous(), 'non-contiguous tensor crossing the C ABI'
    assert t.dtype in (torch.float32, torch.int32), t.dtype
    return t.data_ptr()
