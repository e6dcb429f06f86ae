"""ctypes binding of libtpgan_hip.so (the C-ABI declared in include/tpgan_ops.h).

This is the binding a maintainer of the reference would add in place of the
`pointnet2_ops._ext`, `pytorch3d._C`, `frnn._C` and `chamferdist` extension
modules.  There is NO fallback: if the library is missing the
product path raises, it never routes through a CPU implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TPGAN_HIP_LIBRARY: another build of the same library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("TPGAN_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libtpgan_hip.so")

_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_longlong

# symbol -> argtypes (restype is always int except the two string getters)
SIGNATURES = {
    "tpg_knn_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P],
    "tpg_knn_mfma_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "tpg_knn_grid_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "tpg_frnn_grid_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "tpg_chamfer_fwd_f32": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    "tpg_chamfer_bwd_f32": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "tpg_fps_f32": [_P, _I, _I, _I, _P, _P, _P],
    "tpg_fps_prefix_f32": [_P, _I, _I, _I, _P, _P, _P, _P, _P],
    "tpg_fps_start_f32": [_P, _P, _I, _I, _I, _I, _P, _P, _P],
    "tpg_gather_fwd_f32": [_P, _P, _I, _I, _I, _I, _P, _P],
    "tpg_gather_bwd_f32": [_P, _P, _I, _I, _I, _I, _P, _P],
    "tpg_ball_query_f32": [_P, _P, _I, _I, _I, _F, _I, _P, _P],
    "tpg_group_fwd_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "tpg_group_bwd_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "tpg_three_nn_f32": [_P, _P, _I, _I, _I, _P, _P, _P],
    "tpg_three_interp_fwd_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "tpg_three_interp_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "tpg_gather_rows_fwd_f32": [_P, _P, _I, _I, _I, _I, _P, _P],
    "tpg_gather_rows_bwd_f32": [_P, _P, _I, _I, _I, _I, _P, _P],
    "tpg_rowcombine_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P],
    "tpg_invert_index": [_P, _I, _I, _I, _P, _P, _P, _P],
    "tpg_rowcombine_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P],
    "tpg_rowcombine_edge_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P],
    "tpg_rowcombine_edge_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P],
    "tpg_head_bn_act_fwd": [_P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P],
    "tpg_head_bn_act_bwd": [_P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _P, _P, _P, _P],
    "tpg_rowbn_fwd": [_P, _I, _L, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _P, _P, _I, _I, _P],
    "tpg_rowbn_bwd": [_P, _I, _P, _I, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _P],
    "tpg_cubic_interp_f32": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "tpg_spectral_norm_fwd": [_P, _P, _P, _I, _I, _I, _F, _P, _P, _P],
    "tpg_spectral_norm_bwd": [_P, _P, _P, _P, _P, _I, _I, _P, _P],
    "tpg_spectral_norm_multi_fwd": [_P, _I, _I, _P, _I, _F, _P],
    "tpg_spectral_norm_multi_fwd_split": [_P, _P, _I, _I, _P, _P, _L, _F, _P],
    "tpg_spectral_norm_multi_bwd": [_P, _I, _I, _P, _P, _P, _P, _P],
    "tpg_rowbn_bwd_sums": [_P, _I, _P, _I, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _P],
    "tpg_rowbn_bwd_sums_consts": [_P, _I, _P, _I, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _P],
    "tpg_rowbn_stats_consts": [_P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "tpg_mlp_consts": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "tpg_mlp_max_prep": [_P, _P, _P, _F, _L, _I, _I, _P, _P],
    "tpg_mlp_dgrad": [_P, _P, _P, _I, _P, _P, _P, _F, _P, _I, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "tpg_mlp_wgrad": [_P, _P, _P, _I, _P, _P, _P, _F, _L, _I, _I, _I, _I, _P, _P, _P],
    "tpg_mlp_bn_bwd_apply": [_P, _P, _P, _P, _L, _I, _I, _P, _P],
    "tpg_mlp_bn_bwd_apply_rowsum": [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P],
    "tpg_small_tail_fwd": [_P, _I, _P, _P, _F, _F, _L, _I, _I, _I, _I, _P, _P, _P],
    "tpg_small_tail_bwd": [_P, _P, _P, _P, _I, _P, _P, _F, _F, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "tpg_rowlinear_fwd": [_P, _I, _P, _P, _L, _I, _I, _I, _F, _P, _I, _P],
    "tpg_rowlinear_dgrad": [_P, _P, _I, _P, _L, _I, _I, _I, _F, _P, _I, _P],
    "tpg_rowlinear_wgrad": [_P, _I, _P, _P, _I, _L, _I, _I, _I, _F, _P, _P, _P, _P],
    "tpg_mlp_fwd": [_P, _L, _I, _I, _I, _P, _I, _F, _P, _I, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
}
SIZE_GETTERS = ("tpg_rowbn_workspace_bytes", "tpg_mlp_workspace_bytes")
OTHER_GETTERS = ("tpg_spectral_norm_multi_stride", "tpg_spectral_norm_multi_bwd_scratch",
                 "tpg_mlp_wgrad_workspace_bytes", "tpg_frnn_grid_workspace_bytes", "tpg_small_tail_workspace_bytes",
                 "tpg_chamfer_bwd_workspace_bytes", "tpg_rowlinear_wgrad_workspace_bytes", "tpg_rowlinear_supported",
                 "tpg_spectral_norm_split_rows", "tpg_spectral_norm_split_max_cn", "tpg_spectral_norm_split_max_rows")
STRING_GETTERS = ("tpg_version", "tpg_target_arch")

STATUS = {0: "TPG_OK", -1: "TPG_ERR_ARG", -2: "TPG_ERR_LAUNCH", -3: "TPG_ERR_UNSUPPORTED"}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """dlopen the HIP library (once).  Raises HipLibraryMissing loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name in STRING_GETTERS:
        getattr(lib, name).restype = C.c_char_p
    for name in SIZE_GETTERS:
        getattr(lib, name).argtypes = [C.c_int, C.c_int]
        getattr(lib, name).restype = C.c_size_t
    lib.tpg_spectral_norm_multi_stride.argtypes = [C.c_int, C.c_int]
    lib.tpg_spectral_norm_multi_stride.restype = C.c_longlong
    lib.tpg_spectral_norm_multi_bwd_scratch.argtypes = [C.c_int, C.c_int]
    lib.tpg_spectral_norm_multi_bwd_scratch.restype = C.c_longlong
    lib.tpg_mlp_wgrad_workspace_bytes.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_int]
    lib.tpg_mlp_wgrad_workspace_bytes.restype = C.c_size_t
    lib.tpg_small_tail_workspace_bytes.argtypes = [C.c_longlong, C.c_int]
    lib.tpg_small_tail_workspace_bytes.restype = C.c_size_t
    lib.tpg_frnn_grid_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.tpg_frnn_grid_workspace_bytes.restype = C.c_size_t
    lib.tpg_chamfer_bwd_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.tpg_chamfer_bwd_workspace_bytes.restype = C.c_size_t
    lib.tpg_rowlinear_wgrad_workspace_bytes.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.tpg_rowlinear_wgrad_workspace_bytes.restype = C.c_size_t
    for name in ("tpg_spectral_norm_split_rows", "tpg_spectral_norm_split_max_cn", "tpg_spectral_norm_split_max_rows"):
        getattr(lib, name).argtypes = []
        getattr(lib, name).restype = C.c_int
    lib.tpg_rowlinear_supported.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.tpg_rowlinear_supported.restype = C.c_int
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed: {STATUS.get(rc, rc)}")
