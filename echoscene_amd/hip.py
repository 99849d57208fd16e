"""ctypes binding of libechoscene_hip.so (include/echoscene_hip.h).

The library is the product's only compute path.  ``lib()`` fails loudly when the shared
object is missing -- there is no PyTorch / CPU fallback anywhere in this package.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ES_LIB_TAG: an instrumented build of the same sources (tools/: `_stamp` = phase stamps, echoscene_amd/build.py); never set in product use
LIB_PATH = os.path.join(HERE, 'libechoscene_hip%s.so' % os.environ.get('ES_LIB_TAG', ''))

c_f32p = C.c_void_p
c_i32p = C.c_void_p

# enums (include/echoscene_hip.h)
SEG_DIRECT, SEG_GATHER, SEG_CSRMEAN, SEG_CSRSUM, SEG_CSRWAVG = 0, 1, 2, 3, 4
PRO_NONE, PRO_SILU, PRO_GN, PRO_GN_SILU, PRO_LN, PRO_GEGLU, PRO_LN_ATTN = 0, 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_RELU, ACT_SILU, ACT_GEGLU, ACT_SIGMOID = 0, 1, 2, 3, 4
CONV_SAME, CONV_DOWN_HW, CONV_UP_HW, CONV_UP_DHW, CONV_DOWN_DHW = 0, 1, 2, 3, 4
EPI_NONE, EPI_GEGLU = 0, 1
(OP_LINEAR, OP_DDPM, OP_DDIM, OP_COPY, OP_CONV, OP_GN, OP_LN, OP_ATTN, OP_GEGLU, OP_TO_CL, OP_STEM) = range(1, 12)
OP_VQ = 12
OP_FORK, OP_JOIN, OP_ROWSEL = 13, 14, 15
OP_CONV_F32, OP_ATTN_F32 = 16, 17          # fp32-operand validation route (csrc/es_vol32.hip)


class Seg(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('idx', C.c_void_p), ('ent_row', C.c_void_p), ('ent_off', C.c_void_p),
                ('step', C.c_void_p), ('step_stride', C.c_int32), ('ld', C.c_int32), ('width', C.c_int32),
                ('mode', C.c_int32), ('nslab', C.c_int32), ('slab_stride', C.c_int32), ('pre_act', C.c_int32),
                ('pro', C.c_int32), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('eps', C.c_float), ('gs', C.c_int32),
                ('ent_wt', C.c_void_p)]


class LinearArgs(C.Structure):
    _fields_ = [('seg', Seg * 3), ('nseg', C.c_int32), ('M', C.c_int32), ('K', C.c_int32), ('N', C.c_int32),
                ('wpack', C.c_void_p), ('bias', C.c_void_p), ('prologue', C.c_int32), ('gamma', C.c_void_p),
                ('beta', C.c_void_p), ('eps', C.c_float), ('act', C.c_int32), ('res', C.c_void_p),
                ('res_ld', C.c_int32), ('res_nslab', C.c_int32), ('res_slab_stride', C.c_int32),
                ('res2', C.c_void_p), ('res2_ld', C.c_int32), ('res2_nslab', C.c_int32), ('res2_slab_stride', C.c_int32),
                ('out', C.c_void_p),
                ('out_ld', C.c_int32), ('nbatch', C.c_int32), ('a_bstride', C.c_int32), ('out_bstride', C.c_int32),
                ('kb_per_slice', C.c_int32), ('out_slab_stride', C.c_int32), ('fuse_next', C.c_int32),
                ('res_step', C.c_void_p), ('res_step_stride', C.c_int32), ('seg_slices', C.c_int32)]


class UpdateArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('eps', C.c_void_p), ('eps_nslab', C.c_int32), ('eps_slab_stride', C.c_int32),
                ('noise', C.c_void_p), ('noise_stride', C.c_int32),
                ('coef', C.c_void_p), ('coef_stride', C.c_int32), ('step', C.c_void_p), ('n', C.c_int32),
                ('inc_step', C.c_int32), ('clip_x0', C.c_int32)]


class ConvArgs(C.Structure):
    _fields_ = [('a', C.c_void_p), ('w', C.c_void_p), ('O', C.c_int32), ('D', C.c_int32), ('H', C.c_int32),
                ('W', C.c_int32), ('Cin', C.c_int32), ('N', C.c_int32), ('taps', C.c_int32), ('mode', C.c_int32),
                ('a2', C.c_void_p), ('w2', C.c_void_p), ('Cin2', C.c_int32), ('bias', C.c_void_p),
                ('rowvec', C.c_void_p), ('rowvec_ld', C.c_int32), ('res', C.c_void_p), ('out_f32', C.c_void_p), ('out_f16', C.c_void_p),
                ('workspace', C.c_void_p), ('splitk', C.c_int32), ('out_ld', C.c_int32), ('O_hint', C.c_int32),
                ('epilogue', C.c_int32), ('gn_stats_out', C.c_void_p), ('gn_part_out', C.c_void_p), ('gn_part_groups', C.c_int32)]


class GNArgs(C.Structure):
    _fields_ = [('x1', C.c_void_p), ('C1', C.c_int32), ('x2', C.c_void_p), ('C2', C.c_int32), ('O', C.c_int32),
                ('V', C.c_int32), ('groups', C.c_int32), ('eps', C.c_float), ('gamma', C.c_void_p),
                ('beta', C.c_void_p), ('silu', C.c_int32), ('stats', C.c_void_p), ('y_f16', C.c_void_p),
                ('raw_f16', C.c_void_p), ('O_hint', C.c_int32), ('stats1', C.c_void_p), ('stats2', C.c_void_p), ('x1_is_f16', C.c_int32), ('y_is_f32', C.c_int32), ('part_in', C.c_void_p)]


class LNArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('M', C.c_int32), ('C', C.c_int32), ('eps', C.c_float), ('gamma', C.c_void_p),
                ('beta', C.c_void_p), ('y_f16', C.c_void_p), ('y_is_f32', C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [('qkv', C.c_void_p), ('B', C.c_int32), ('Ntok', C.c_int32), ('heads', C.c_int32),
                ('dhead', C.c_int32), ('scale', C.c_float), ('out_f16', C.c_void_p)]


class GegluArgs(C.Structure):
    _fields_ = [('h_f32', C.c_void_p), ('M', C.c_int32), ('C4', C.c_int32), ('out_f16', C.c_void_p), ('out_is_f32', C.c_int32)]


class RowSelArgs(C.Structure):
    _fields_ = [('table', C.c_void_p), ('stride', C.c_int32), ('step', C.c_void_p), ('out', C.c_void_p),
                ('out_ld', C.c_int32), ('rows', C.c_int32), ('n', C.c_int32)]


class CopyArgs(C.Structure):
    _fields_ = [('dst', C.c_void_p), ('src', C.c_void_p), ('bytes', C.c_size_t), ('rows', C.c_int32),
                ('dst_pitch', C.c_size_t), ('src_pitch', C.c_size_t)]


class ToClArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('O', C.c_int32), ('C', C.c_int32), ('V', C.c_int32), ('Cpad', C.c_int32),
                ('out', C.c_void_p), ('out_is_f32', C.c_int32)]


class StemArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w0', C.c_void_p), ('b0', C.c_void_p), ('w1', C.c_void_p), ('b1', C.c_void_p),
                ('scratch', C.c_void_p), ('out', C.c_void_p), ('O', C.c_int32), ('Cin', C.c_int32), ('x_ostride', C.c_int32)]


class VQArgs(C.Structure):
    _fields_ = [('z', C.c_void_p), ('codebook', C.c_void_p), ('lut', C.c_void_p), ('O', C.c_int32), ('V', C.c_int32),
                ('n_embed', C.c_int32), ('Cpad', C.c_int32), ('idx_out', C.c_void_p), ('out_f16', C.c_void_p)]


class _OpU(C.Union):
    _fields_ = [('linear', LinearArgs), ('update', UpdateArgs), ('copy', CopyArgs), ('conv', ConvArgs),
                ('gn', GNArgs), ('ln', LNArgs), ('attn', AttnArgs), ('geglu', GegluArgs), ('tocl', ToClArgs),
                ('stem', StemArgs), ('vq', VQArgs), ('rowsel', RowSelArgs)]


class Op(C.Structure):
    _fields_ = [('kind', C.c_int32), ('lane', C.c_int32), ('u', _OpU)]


class BufferDesc(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('bytes', C.c_size_t)]


class RegionDesc(C.Structure):
    _fields_ = [('name', C.c_char * 32), ('ptr', C.c_void_p), ('bytes', C.c_size_t)]


EXPORTS = {
    'es_abi_version': (C.c_int, []),
    'es_last_error': (C.c_char_p, []),
    'es_device_info': (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    'es_pack_linear_f32_size': (C.c_size_t, [C.c_int, C.c_int]),
    'es_pack_linear_f32': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'es_pack_linear_f32_dev': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'es_matmul_f64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'es_pack_linear_geglu_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'es_linear_rows_f32': (C.c_int, [C.POINTER(LinearArgs), C.c_void_p]),
    'es_linear_rows_multi_f32': (C.c_int, [C.POINTER(C.POINTER(LinearArgs)), C.c_int, C.c_void_p]),
    'es_rows_set_kernel_family': (C.c_int, [C.c_int]),
    'es_rows_get_kernel_family': (C.c_int, []),
    'es_vol_set_option': (C.c_int, [C.c_char_p, C.c_int]),
    'es_vol_options': (C.c_int, [C.c_char_p, C.c_int]),
    'es_options_string': (C.c_int, [C.c_char_p, C.c_int]),
    'es_model_file_options': (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    'es_linear_rows_slices': (C.c_int, [C.POINTER(LinearArgs), C.POINTER(C.c_int)]),
    'es_linear_rows_takes_ln_attn': (C.c_int, [C.POINTER(LinearArgs)]),
    'es_linear_rows_auto_slices': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'es_row_select': (C.c_int, [C.POINTER(RowSelArgs), C.c_void_p]),
    'es_ddpm_update': (C.c_int, [C.POINTER(UpdateArgs), C.c_void_p]),
    'es_ddim_update': (C.c_int, [C.POINTER(UpdateArgs), C.c_void_p]),
    'es_box_postprocess': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                     C.c_void_p]),
    'es_box_descale': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'es_conv_mfma_f16': (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    'es_conv_emits_gn_stats': (C.c_int, [C.POINTER(ConvArgs)]),
    'es_conv_emits_gn_part': (C.c_int, [C.POINTER(ConvArgs)]),
    'es_conv_split_of': (C.c_int, [C.POINTER(ConvArgs)]),
    'es_split_f16x3': (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    'es_pack_conv_f16_size': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'es_pack_conv_f16': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'es_pack_conv_f16_dev': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'es_pack_conv_rows_f16': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'es_groupnorm_vol': (C.c_int, [C.POINTER(GNArgs), C.c_void_p]),
    'es_layernorm_tokens': (C.c_int, [C.POINTER(LNArgs), C.c_void_p]),
    'es_attention_f16': (C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    'es_geglu_f16': (C.c_int, [C.POINTER(GegluArgs), C.c_void_p]),
    'es_latent_to_cl_f16': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'es_latent_to_cl_f32': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'es_pack_conv_f32_size': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'es_pack_conv_f32': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'es_conv_f32': (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    'es_attention_f32': (C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    'es_shape_stem': (C.c_int, [C.POINTER(StemArgs), C.c_void_p]),
    'es_vq_lookup': (C.c_int, [C.POINTER(VQArgs), C.c_void_p]),
    'es_init': (C.c_int, []),
    'es_chamfer_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    'es_chamfer_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'es_marching_cubes_workspace': (C.c_size_t, [C.c_int, C.c_int]),
    'es_marching_cubes_count': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    'es_marching_cubes_emit': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'es_plan_create': (C.c_void_p, [C.POINTER(Op), C.c_int]),
    'es_plan_destroy': (None, [C.c_void_p]),
    'es_plan_num_ops': (C.c_int, [C.c_void_p]),
    'es_plan_run': (C.c_int, [C.c_void_p, C.c_void_p]),
    'es_plan_capture': (C.c_int, [C.c_void_p, C.c_void_p]),
    'es_sampler_run': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'es_op_pointer_offsets': (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.c_int]),
    'es_model_save': (C.c_int, [C.c_char_p, C.c_void_p, C.POINTER(BufferDesc), C.c_int, C.POINTER(RegionDesc), C.c_int]),
    'es_model_load': (C.c_void_p, [C.c_char_p]),
    'es_model_free': (None, [C.c_void_p]),
    'es_model_region': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    'es_model_num_ops': (C.c_int, [C.c_void_p]),
    'es_model_run': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'es_layout_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'es_shape_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'es_vq_decode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  -- torch's bundled HIP runtime must be the one the process binds first
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libechoscene_hip.so is missing (%s). Build it with `python -m echoscene_amd.build` '
                '(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback.' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)         # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if L.es_abi_version() != 10:
            raise RuntimeError('libechoscene_hip.so ABI version mismatch')
        _lib = L
    return _lib


def check(rc, what=''):
    if rc != 0:
        raise RuntimeError('%s failed (status %d): %s' % (what or 'echoscene_hip call', rc,
                                                          lib().es_last_error().decode()))


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())
