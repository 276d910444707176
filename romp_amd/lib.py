"""ctypes binding of libromp_hip.so (the C ABI declared in include/romp_hip.h).

PyTorch-ROCm is plumbing here: it owns device memory and streams; every arithmetic step of
the hot path is a call into the HIP library.  The library is REQUIRED -- there is no CPU
or eager-PyTorch fallback in the product path; a missing/unbuildable extension raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ROMP_HIP_LIB') or os.path.join(_HERE, 'libromp_hip.so')     # (ROMP_HIP_LIB: a debug build of the same ABI)

ABI_VERSION = 7          # ROMP_ABI_VERSION of include/romp_hip.h this binding was written against
BUF_NONE, BUF_IMAGE, BUF_CENTER, BUF_PARAMS = -1, -2, -3, -4
FMT_F32, FMT_H2 = 0, 1
OP_STEM, OP_CONV, OP_FUSESUM, OP_FORK, OP_JOIN, OP_BEV_PACK, OP_BEV_MAPS, OP_CONV3D = 1, 2, 3, 4, 5, 6, 7, 8
OP_NOP, OP_BBLOCK32, OP_BBLOCK64, OP_SEAM1X1, OP_FUSEUP, OP_RECORD, OP_WAIT = 12, 13, 14, 15, 16, 17, 18
OP_KSUM = 11
OP_STEM2 = 19
OP_STEM7, OP_MAXPOOL, OP_STEM7P = 9, 10, 20
OPF_WAVE16, OPF_STEM_VALU, OPF_SEAM_DS = 1, 2, 4


class RompOp(C.Structure):
    """Mirror of `struct romp_op` (include/romp_hip.h)."""
    _fields_ = [
        ('kind', C.c_int32),
        ('in_buf', C.c_int32), ('out_buf', C.c_int32), ('res_buf', C.c_int32),
        ('H', C.c_int32), ('W', C.c_int32),
        ('Cin', C.c_int32), ('Cout', C.c_int32),
        ('ksize', C.c_int32), ('stride', C.c_int32), ('relu', C.c_int32),
        ('groups', C.c_int32),
        ('in_cstride', C.c_int32), ('in_coff', C.c_int32), ('in_gstride', C.c_int32),
        ('out_cstride', C.c_int32), ('out_coff', C.c_int32), ('out_gstride', C.c_int32),
        ('res_cstride', C.c_int32), ('res_coff', C.c_int32), ('res_gstride', C.c_int32),
        ('cin_pad', C.c_int32), ('cout_pad', C.c_int32),
        ('n_terms', C.c_int32),
        ('term_buf', C.c_int32 * 4), ('term_shift', C.c_int32 * 4), ('term_cstride', C.c_int32 * 4),
        ('stream', C.c_int32), ('pad_h', C.c_int32), ('pad_w', C.c_int32), ('out_rstride', C.c_int32), ('out_bstride', C.c_int32),
        ('in_fmt', C.c_int32), ('out_fmt', C.c_int32), ('res_fmt', C.c_int32), ('term_fmt', C.c_int32 * 4),
        ('act_shift', C.c_int32),
        ('flags', C.c_int32), ('relu_from', C.c_int32), ('term_coff', C.c_int32 * 4),
        ('weight', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p), ('weight_aux', C.c_void_p),
        ('weight_h2', C.c_void_p), ('scale_h2', C.c_void_p),
    ]


_lib = None


class RompHipError(RuntimeError):
    pass


def load():
    """Load libromp_hip.so (building it in-tree with hipcc if it is missing).  torch must be
    imported first so that the library binds to the same libamdhip64 as PyTorch-ROCm."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's libamdhip64.so.7 before ours resolves it)
    if not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64p, f = C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_float
    sigs = {
        'romp_abi_version': (C.c_int, []),
        'romp_last_error': (C.c_char_p, []),
        'romp_net_create': (C.c_int, [C.POINTER(vp), C.POINTER(RompOp), i32, i64p, i32, i32]),
        'romp_net_forward': (C.c_int, [vp, vp, i32, vp, vp, vp]),
        'romp_net_read_buffer': (C.c_int, [vp, i32, i32, vp, C.c_int64, vp]),
        'romp_net_write_buffer': (C.c_int, [vp, i32, vp, C.c_int64, vp]),
        'romp_net_set_mode': (C.c_int, [vp, i32]),
        'romp_net_set_graph': (C.c_int, [vp, i32]),
        'romp_net_set_streams': (C.c_int, [vp, i32]),
        'romp_net_profile': (C.c_int, [vp, vp, i32, vp, vp, vp, C.POINTER(C.c_float), i32]),
        'romp_net_range_scan': (C.c_int, [vp, vp, i32, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        'romp_net_saturated': (C.c_int, [vp, C.POINTER(C.c_int64), i32, vp]),
        'romp_net_set_sat_check': (C.c_int, [vp, i32]),
        'romp_net_destroy': (None, [vp]),
        'romp_conv_forward': (C.c_int, [C.POINTER(RompOp), vp, vp, vp, i32, i32, i32, vp]),
        'romp_conv_num_variants': (C.c_int, []),
        'romp_conv_family_variants': (C.c_int, [i32]),
        'romp_conv_trace_read': (C.c_int, [C.POINTER(C.c_uint64), i32]),
        'romp_conv_describe': (C.c_int, [C.POINTER(RompOp), i32, i32, C.c_char_p, i32]),
        'romp_net_autotune': (C.c_int, [vp, i32, i32, vp]),
        'romp_net_tuned_variant': (C.c_int, [vp, i32, i32]),
        'romp_oneeuro_state_floats': (C.c_int, [i32]),
        'romp_oneeuro_smooth': (C.c_int, [vp, vp, i32, i32, f, vp, vp, vp, vp]),
        'romp_cam_to_trans': (C.c_int, [vp, i32, f, vp, vp]),
        'romp_estimate_translation': (C.c_int, [vp, i32, i32, i32, vp, f, f, vp, vp]),
        'romp_project_verts': (C.c_int, [vp, i32, i32, vp, C.POINTER(C.c_float), vp, vp, vp]),
        'romp_bev_project_verts': (C.c_int, [vp, i32, i32, vp, C.POINTER(C.c_float), vp, vp]),
        'romp_sim3dr_normals': (C.c_int, [vp, vp, vp, vp, i32, vp, vp]),
        'romp_sim3dr_light': (C.c_int, [vp, vp, i32, C.POINTER(C.c_float), vp, vp]),
        'romp_sim3dr_rasterize': (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
        'romp_net_load': (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, i32]),
        'romp_net_plan_info': (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
        'romp_net_plan_kind': (C.c_int, [vp, C.POINTER(C.c_int32)]),
        'romp_net_set_split': (C.c_int, [vp, i32, i32, C.c_int64, C.c_int64, C.c_int64]),
        'romp_net_set_tuned': (C.c_int, [vp, i32, C.POINTER(C.c_int32), i32]),
        'romp_bev_workspace_ints': (C.c_int, [i32, i32]),
        'romp_bev_parse': (C.c_int, [vp, i32, f, i32, C.POINTER(C.c_int32), vp, vp, vp, vp, vp]),
        'romp_bev_regress': (C.c_int, [vp, vp, i32, i32, vp, vp, C.POINTER(C.c_float)] + [vp] * 7 + [vp] * 6 + [vp]),
        'romp_net_buffer_ptr': (C.c_void_p, [vp, i32]),
        'romp_parse': (C.c_int, [vp, vp, i32, f, i32, C.POINTER(C.c_int32), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        'romp_parse_watch': (C.c_int, [vp, vp, i32, f, i32, C.POINTER(C.c_int32), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int32)]),
        'romp_net_sat_counter': (C.c_void_p, [vp]),
        'romp_rot6d_to_aa': (C.c_int, [vp, i32, vp, vp]),
        'smpl_ctx_create': (C.c_int, [C.POINTER(vp), vp, vp, i32, vp, vp, vp, i64p, vp, vp, i64p, i32, vp]),
        'smpl_forward': (C.c_int, [vp, vp, i32, vp, i32, i32, vp, vp, vp]),
        'smpl_ctx_destroy': (None, [vp]),
        'romp_preprocess': (C.c_int, [vp, i32, i32, vp, i32, C.POINTER(C.c_float), vp]),
        'romp_preprocess_batch': (C.c_int, [vp, i32, i32, i32, vp, i32, C.POINTER(C.c_float), vp]),
        'romp_bev_postprocess': (C.c_int, [vp, vp, vp, i32, vp, f, f, vp, vp, vp, vp, vp]),
        'romp_project': (C.c_int, [vp, i32, i32, vp, C.POINTER(C.c_float), vp, vp, vp, vp]),
    }
    # the version first: a stale or mismatched library must fail with THIS message, not with a missing-symbol AttributeError
    lib.romp_abi_version.restype = C.c_int
    if lib.romp_abi_version() != ABI_VERSION:
        raise RompHipError('%s is ABI version %d, this binding needs %d: rebuild it (python -m romp_amd.build --force)'
                           % (LIB_PATH, lib.romp_abi_version(), ABI_VERSION))
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)      # raises AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args

    _lib = lib
    return lib


EXPORTS = ['romp_abi_version', 'romp_last_error', 'romp_net_create', 'romp_net_forward', 'romp_net_read_buffer',
           'romp_net_write_buffer', 'romp_net_set_mode', 'romp_net_set_graph', 'romp_net_set_streams', 'romp_net_profile', 'romp_net_range_scan', 'romp_net_saturated', 'romp_net_set_sat_check', 'romp_net_destroy',
           'romp_conv_forward', 'romp_conv_num_variants', 'romp_conv_family_variants', 'romp_conv_trace_read', 'romp_conv_describe',
           'romp_net_load', 'romp_net_plan_info', 'romp_net_plan_kind', 'romp_net_autotune', 'romp_net_tuned_variant', 'romp_net_set_tuned', 'romp_net_set_split', 'romp_project_verts', 'romp_estimate_translation', 'romp_cam_to_trans', 'romp_bev_project_verts', 'romp_oneeuro_state_floats', 'romp_oneeuro_smooth', 'romp_sim3dr_normals', 'romp_sim3dr_light', 'romp_sim3dr_rasterize', 'romp_bev_workspace_ints', 'romp_bev_parse', 'romp_bev_regress',
           'romp_net_buffer_ptr', 'romp_parse', 'romp_parse_watch', 'romp_net_sat_counter', 'romp_rot6d_to_aa', 'smpl_ctx_create', 'smpl_forward', 'smpl_ctx_destroy',
           'romp_project', 'romp_preprocess', 'romp_preprocess_batch', 'romp_bev_postprocess']


def has_bf16x3():
    """Does this build of the library carry the optional bf16x3 kernel family (ROMP_WITH_BX3=1 python -m romp_amd.build)?"""
    h = load()
    return h.romp_conv_family_variants(1) + h.romp_conv_family_variants(2) > 0


def check(rc):
    if rc != 0:
        msg = load().romp_last_error()
        raise RompHipError('libromp_hip error %d: %s' % (rc, msg.decode() if msg else '?'))


def ptr(t):
    """Device pointer of a torch tensor as c_void_p (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
