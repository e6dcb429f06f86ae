"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares;
the product path refuses CPU tensors (no fallback) and never imports the oracle."""
import glob
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(tpg_[a-z0-9_]+)\s*\(", txt))
    return names


def test_header_declares_the_whole_boundary():
    names = _declared_symbols()
    for need in ["tpg_knn_f32", "tpg_chamfer_fwd_f32", "tpg_chamfer_bwd_f32", "tpg_fps_f32",
                 "tpg_gather_fwd_f32", "tpg_gather_bwd_f32", "tpg_ball_query_f32",
                 "tpg_group_fwd_f32", "tpg_group_bwd_f32", "tpg_three_nn_f32",
                 "tpg_three_interp_fwd_f32", "tpg_three_interp_bwd_f32"]:
        assert need in names


def test_library_exports_every_declared_symbol(hip_lib):
    from tpgan_amd import _lib
    for name in _declared_symbols():
        assert hasattr(hip_lib, name), f"{name} declared in include/ but not exported"
    assert set(_lib.SIGNATURES) | set(_lib.STRING_GETTERS) | set(_lib.SIZE_GETTERS) | set(_lib.OTHER_GETTERS) == _declared_symbols()
    assert hip_lib.tpg_target_arch() == b"gfx950"


def test_code_object_targets_gfx950(hip_lib):
    from tpgan_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_cpu_tensors_are_refused_without_a_registered_checker():
    import tpgan_amd.ops as ops
    ops.unregister_backend("cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.furthest_point_sample(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.grouping_operation(torch.zeros(1, 2, 8), torch.zeros(1, 4, 2, dtype=torch.int32))


def test_product_package_never_imports_the_oracle():
    code = ("import sys; sys.path.insert(0, %r); import tpgan_amd, tpgan_amd.ops; "
            "tpgan_amd.install_compat(); import pointnet2_ops.pointnet2_utils, pytorch3d.ops, frnn, "
            "chamferdist; import tpgan_amd.gan_step; "
            "bad=[m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; "
            "assert not bad, bad") % ROOT
    subprocess.check_call([sys.executable, "-c", code])
    pkg = os.path.join(ROOT, "temporal-pointcloud-upsampling-gan_amd")
    for path in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path


def test_validation_errors_match_upstream_style(oracle_cpu):
    import tpgan_amd.ops as ops
    f = torch.zeros(1, 2, 8)
    with pytest.raises(RuntimeError, match="int tensor"):
        ops.grouping_operation(f, torch.zeros(1, 4, 2, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="contiguous"):
        ops.grouping_operation(f.transpose(1, 2).contiguous().transpose(1, 2),
                               torch.zeros(1, 4, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="float tensor"):
        ops.furthest_point_sample(torch.zeros(1, 8, 3, dtype=torch.float64), 4)


def test_new_entries_reject_bad_arguments_before_any_launch(hip_lib):
    """Argument checks of the round-2 entries return their status codes without touching a device (no GPU here):
    unsupported shapes are TPG_ERR_UNSUPPORTED (-3), malformed calls TPG_ERR_ARG (-1), empty work TPG_OK."""
    import ctypes as C
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    # feature-space kNN filter: D in {32, 64}, 2 <= K <= 24
    assert hip_lib.tpg_knn_mfma_f32(p, p, None, None, 1, 8, 8, 48, 4, p, p, 1, None) == -3
    assert hip_lib.tpg_knn_mfma_f32(p, p, None, None, 1, 8, 8, 32, 25, p, p, 1, None) == -3
    assert hip_lib.tpg_knn_mfma_f32(p, p, None, None, 1, 8, 8, 32, 1, p, p, 1, None) == -1
    assert hip_lib.tpg_knn_mfma_f32(p, p, None, None, 0, 8, 8, 32, 4, p, p, 1, None) == 0
    # grid searches: K <= 64, workspace required
    assert hip_lib.tpg_knn_grid_f32(p, p, None, None, 1, 8, 8, 65, p, p, p, None) == -1
    assert hip_lib.tpg_knn_grid_f32(p, p, None, None, 1, 8, 8, 4, p, p, None, None) == -1
    assert hip_lib.tpg_knn_grid_f32(p, p, None, None, 1, 0, 8, 4, p, p, p, None) == 0
    assert hip_lib.tpg_frnn_grid_f32(p, p, None, None, 1, 8, 8, 4, 0.0, p, p, p, None) == -1
    # the 16 -> 16 -> 32 tails: other channel counts are not built
    assert hip_lib.tpg_small_tail_fwd(p, 0, p, p, 0.2, 0.2, 16, 4, 32, 16, 32, p, p, None) == -3
    assert hip_lib.tpg_small_tail_fwd(p, 0, p, p, 0.2, 0.2, 16, 0, 16, 16, 32, p, p, None) == -1
    assert hip_lib.tpg_small_tail_fwd(p, 0, p, p, 0.2, 0.2, 0, 4, 16, 16, 32, p, p, None) == 0
    assert hip_lib.tpg_small_tail_workspace_bytes(12288, 20) > 0 and hip_lib.tpg_frnn_grid_workspace_bytes(0, 8) == 0
    # BatchNorm backward + row sums: whole groups of K rows per segment, channel chunks of 8
    assert hip_lib.tpg_mlp_bn_bwd_apply_rowsum(p, p, p, p, 64, 5, 64, 1, p, p, None) == -1       # 64 rows, K = 5
    assert hip_lib.tpg_mlp_bn_bwd_apply_rowsum(p, p, p, p, 64, 0, 64, 1, p, p, None) == -1
    assert hip_lib.tpg_mlp_bn_bwd_apply_rowsum(p, p, p, p, 64, 8, 64, 1, p, None, None) == -1
    assert hip_lib.tpg_mlp_bn_bwd_apply_rowsum(p, p, p, p, 64, 8, 60, 1, p, p, None) == -3
    # EdgeConv front end on one product (round 3): the node half must start 16-byte aligned, both halves present
    assert hip_lib.tpg_rowcombine_edge_fwd(p, p, 0, 0, 1, 8, 4, 6, 0.2, 0.2, p, None) == -3      # C = 6 floats
    assert hip_lib.tpg_rowcombine_edge_fwd(None, p, 0, 0, 1, 8, 4, 8, 0.2, 0.2, p, None) == -1
    assert hip_lib.tpg_rowcombine_edge_fwd(p, p, 0, 0, 0, 8, 4, 8, 0.2, 0.2, p, None) == 0
    assert hip_lib.tpg_rowcombine_edge_fwd(p, p, 2, 0, 1, 8, 4, 8, 0.2, 0.2, p, None) == -3      # unknown dtype
    assert hip_lib.tpg_rowcombine_edge_bwd(p, p, p, p, None, 0, 0, 1, 8, 4, 8, 0.2, 0.2, p, None) == -1
    assert hip_lib.tpg_rowcombine_edge_bwd(p, p, p, p, p, 0, 0, 1, 8, 4, 8, 0.2, 0.2, None, None) == -1
    assert hip_lib.tpg_rowcombine_edge_bwd(p, p, p, p, p, 0, 0, 1, 8, 4, 6, 0.2, 0.2, p, None) == -3
    # split spectral norm: descriptor, part map and a 16-byte aligned exchange buffer required; <= 576 columns
    assert hip_lib.tpg_spectral_norm_multi_fwd_split(None, p, 1, 64, p, p, 8, 1e-12, None) == -1
    assert hip_lib.tpg_spectral_norm_multi_fwd_split(p, p, 1, 64, p, None, 8, 1e-12, None) == -1
    assert hip_lib.tpg_spectral_norm_multi_fwd_split(p, p, 0, 64, p, p, 8, 1e-12, None) == 0
    assert hip_lib.tpg_spectral_norm_multi_fwd_split(p, p, 1, 577, p, p, 8, 1e-12, None) == -3
    assert hip_lib.tpg_spectral_norm_split_rows() == 32 and hip_lib.tpg_spectral_norm_split_max_cn() == 576
    # rows gather
    assert hip_lib.tpg_gather_rows_fwd_f32(p, p, 1, 0, 4, 3, p, None) == -1
    assert hip_lib.tpg_gather_rows_fwd_f32(None, p, 1, 8, 4, 3, p, None) == -1
    assert hip_lib.tpg_gather_rows_fwd_f32(p, p, 0, 8, 4, 3, p, None) == 0
    assert hip_lib.tpg_gather_rows_bwd_f32(p, p, 1, 8, 4, 3, None, None) == -1
    # head BatchNorm1d + LeakyReLU + mask: one row is an error (as nn.BatchNorm1d in training mode), outputs required
    assert hip_lib.tpg_head_bn_act_fwd(p, 1, 8, p, p, p, p, None, 0.1, 1e-5, 0.01, None, p, p, p, None) == -1
    assert hip_lib.tpg_head_bn_act_fwd(p, 4, 8, p, p, p, p, None, 0.1, 1e-5, 0.01, None, None, p, p, None) == -1
    assert hip_lib.tpg_head_bn_act_fwd(p, 0, 8, p, p, p, p, None, 0.1, 1e-5, 0.01, None, p, p, p, None) == 0
    assert hip_lib.tpg_head_bn_act_bwd(p, p, p, p, p, p, 0.01, None, 4, 8, None, p, p, None) == -1
    assert hip_lib.tpg_head_bn_act_bwd(p, p, p, p, p, p, 0.01, None, 4, 0, p, p, p, None) == 0
    # statistics / backward sums with the folded constants from the same finalize launch
    assert hip_lib.tpg_rowbn_stats_consts(p, 1, 64, 64, 1e-5, 0.1, None, None, None, None, p, p, p, p, None, p, 1, None) == -1
    assert hip_lib.tpg_rowbn_stats_consts(p, 1, 64, 60, 1e-5, 0.1, None, None, None, None, p, p, p, p, p, p, 1, None) == -3
    assert hip_lib.tpg_rowbn_bwd_sums_consts(p, 1, p, 1, p, p, 1, 64, 8, 64, 1, p, p, p, p, 0.2, None, None, p, None, None,
                                             p, 1, None) == -1      # no cb
    assert hip_lib.tpg_rowbn_bwd_sums_consts(p, 1, p, 1, p, None, 1, 64, 8, 64, 1, p, p, p, p, 0.2, None, None, p, p, p,
                                             p, 1, None) == -3      # ag wanted, no y to take lrelu' from
