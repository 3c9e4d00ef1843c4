"""CPU checks of the drop-in boundary: libgcpx.so loads, exports every symbol include/gcpx.h declares, and the
ctypes structs have the same layout as the C structs (no compute calls — there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gcpx.h")


@pytest.fixture(scope="module")
def lib():
    from video_gcp_amd import runtime
    if not os.path.exists(runtime.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return runtime.load_library()


def test_exports_every_declared_symbol(lib):
    from video_gcp_amd import runtime
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(gcpx_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations parsed"
    bound = {name for name, _, _ in runtime.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.gcpx_version() == 1


def test_struct_layout_matches_header():
    from video_gcp_amd import runtime as rt
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "gcpx.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu\n", sizeof(gcpx_conv_src), sizeof(gcpx_conv_args), sizeof(gcpx_row_src),
             sizeof(gcpx_gemm_args), sizeof(gcpx_mlp_args));
      printf("%zu %zu %zu %zu\n", offsetof(gcpx_conv_args, wpk), offsetof(gcpx_conv_args, stats_partial),
             offsetof(gcpx_gemm_args, wpk), offsetof(gcpx_gemm_args, h_copy));
      printf("%zu %zu %zu %zu\n", offsetof(gcpx_mlp_args, w_in), offsetof(gcpx_mlp_args, gn_eps),
             offsetof(gcpx_mlp_args, out), offsetof(gcpx_mlp_args, zrow));
      printf("%zu %zu %zu %zu %zu\n", sizeof(gcpx_wgrad_args), sizeof(gcpx_lstm_bwd_args), sizeof(gcpx_tree_accum_args),
             sizeof(gcpx_actbwd_args), sizeof(gcpx_loss_args));
      printf("%zu %zu %zu %zu %zu\n", offsetof(gcpx_wgrad_args, ldy), offsetof(gcpx_wgrad_args, z_bias_off),
             offsetof(gcpx_lstm_bwd_args, dcp_stride), offsetof(gcpx_tree_accum_args, dst), offsetof(gcpx_actbwd_args, ldc));
      printf("%zu %zu %zu\n", offsetof(gcpx_conv_args, src_row_map), offsetof(gcpx_gemm_args, gates_out),
             offsetof(gcpx_mlp_args, save));
      return 0;
    }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    got = [int(x) for x in out]
    want = [C.sizeof(rt.ConvSrc), C.sizeof(rt.ConvArgs), C.sizeof(rt.RowSrc), C.sizeof(rt.GemmArgs), C.sizeof(rt.MlpArgs),
            rt.ConvArgs.wpk.offset, rt.ConvArgs.stats_partial.offset, rt.GemmArgs.wpk.offset, rt.GemmArgs.h_copy.offset,
            rt.MlpArgs.w_in.offset, rt.MlpArgs.gn_eps.offset, rt.MlpArgs.out.offset, rt.MlpArgs.zrow.offset,
            C.sizeof(rt.WgradArgs), C.sizeof(rt.LstmBwdArgs), C.sizeof(rt.TreeAccumArgs), C.sizeof(rt.ActBwdArgs),
            C.sizeof(rt.LossArgs), rt.WgradArgs.ldy.offset, rt.WgradArgs.z_bias_off.offset, rt.LstmBwdArgs.dcp_stride.offset,
            rt.TreeAccumArgs.dst.offset, rt.ActBwdArgs.ldc.offset, rt.ConvArgs.src_row_map.offset,
            rt.GemmArgs.gates_out.offset, rt.MlpArgs.save.offset]
    assert got == want, (got, want)


def test_missing_library_fails_loudly(tmp_path):
    from video_gcp_amd import runtime as rt
    with pytest.raises(rt.GcpxError):
        rt.load_library(str(tmp_path / "nope.so"))


def test_invalid_args_are_rejected_without_a_gpu(lib):
    """argument validation happens before any HIP call"""
    from video_gcp_amd import runtime as rt
    a = rt.GemmArgs()
    assert lib.gcpx_gemm(C.byref(a), None) == -1
    assert b"nsrc" in lib.gcpx_last_error()
    m = rt.MlpArgs()
    assert lib.gcpx_mlp(C.byref(m), None) == -1
    c = rt.ConvArgs()
    assert lib.gcpx_conv3x3(C.byref(c), None) == -1
    w = rt.WgradArgs()
    assert lib.gcpx_wgrad(C.byref(w), None) == -1
    assert lib.gcpx_lstm_bwd(C.byref(rt.LstmBwdArgs()), None) == -1
    assert lib.gcpx_act_bwd(C.byref(rt.ActBwdArgs()), None) == -1
    assert lib.gcpx_radam_step(None, None, None, None, None, 0, 0.0, 0.9, 0.999, 1e-8, 1.0, None) == -1
    # round-4 entry points: the optimizer slice, the block-limited re-pack, the grouped Predictor backward, the two-launch split re-pack
    assert lib.gcpx_optim_range(None, None, None, None, None, 0, 0, 0.0, 0.9, 0.999, 1e-8, 1.0, 1, 0, None) == -1
    assert lib.gcpx_repack_blocks(None, None, None, None, 0, 0, None) == -1
    assert lib.gcpx_mlp_bwd_group(None, 0, None) == -1
    assert b"GCPX_MLP_BWD_GROUP_MAX" in lib.gcpx_last_error()
    tab = (rt.MlpBwdArgs * 5)()
    assert lib.gcpx_mlp_bwd_group(tab, 5, None) == -1                  # more than GCPX_MLP_BWD_GROUP_MAX problems
    assert lib.gcpx_mlp_bwd_group(tab, 2, None) == -1                  # empty descriptors: rejected before any launch
    assert lib.gcpx_split_pack_group2(None, 0, None, None) == -1
    # round-5 entry points: the collectives (argument checks come before RCCL is even looked for)
    assert lib.gcpx_comm_unique_id(None) == -1
    comm = C.c_void_p()
    assert lib.gcpx_comm_init(C.byref(comm), 2, 2, None) == -1 and lib.gcpx_comm_init(C.byref(comm), 2, 2, C.create_string_buffer(128)) == -1
    assert lib.gcpx_comm_allreduce(None, None, 0, None) == -1 and lib.gcpx_comm_allgather(None, None, None, 0, None) == -1
    assert lib.gcpx_comm_destroy(None) == -1
