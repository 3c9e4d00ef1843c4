"""CPU (hipcc cross-compiles): the output head and the likelihood kernels are built without packed-f32 VALU instructions
(csrc/build.sh: -fno-slp-vectorize for conv3x3_head_split.hip and loss.hip; profiles/r05_head_store_hazard.txt: every wrong value of the
round-4 gradient-row corruption came out of a packed multiply whose low lane read the high half of a register pair)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_head_and_loss_kernels_carry_no_packed_f32():
    env = dict(os.environ, PATH="/opt/rocm/bin:" + os.environ.get("PATH", ""))
    files = [os.path.join(ROOT, "video-gcp_amd", "csrc", f) for f in ("conv3x3_head_split.hip", "loss.hip")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazard_scan.py")] + files, capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for f in ("conv3x3_head_split.hip", "loss.hip"):
        line = [l for l in r.stdout.splitlines() if l.startswith(f"== {f}:")]
        assert line and ": 0 packed-f32 VALU instructions, 0 with a low lane reading a high half" in line[0], r.stdout[-2000:]
    # the per-file flag has ONE home (csrc/sources.sh) that the build, the variant builds and the scan all read
    src = open(os.path.join(ROOT, "video-gcp_amd", "csrc", "sources.sh")).read()
    assert 'GCPX_NO_SLP="conv3x3_head_split conv3x3_head32 loss scalar_f32"' in src and "-fno-slp-vectorize" in src
    for p in ("video-gcp_amd/csrc/build.sh", "tools/build_variant.sh"):
        assert "sources.sh" in open(os.path.join(ROOT, p)).read(), p
    assert "sources.sh" in open(os.path.join(ROOT, "tools", "isa_hazard_scan.py")).read()


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_kernel_of_the_library_has_a_packed_f32_low_lane_reading_a_high_half():
    """the instruction form every wrong value of the round-4 corruption came out of (a v_pk_*_f32 with an op_sel bit set on a VGPR
    source) appears NOWHERE in the library: the five kernels that compiled to it (optimizers, likelihood / mixture-mean backward) live in
    scalar_f32.hip, built without SLP vectorisation, and that file holds no packed-f32 instruction at all"""
    env = dict(os.environ, PATH="/opt/rocm/bin:" + os.environ.get("PATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazard_scan.py")], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "packed-f32 instructions whose low lane reads a high half: 0;" in r.stdout, r.stdout[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("== scalar_f32.hip:")]
    assert line and ": 0 packed-f32 VALU instructions" in line[0], r.stdout[-2000:]
