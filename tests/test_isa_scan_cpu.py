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
    # the per-file flag lives in three places that must agree: the build, the variant builds and the scan
    for p, needle in (("video-gcp_amd/csrc/build.sh", "-fno-slp-vectorize"), ("tools/build_variant.sh", "-fno-slp-vectorize")):
        assert needle in open(os.path.join(ROOT, p)).read(), p
