"""Layout probe for the output-head data gradient: 112-channel interleaved source vs the same bytes as 16-channel frames."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_gcp_amd as V
from video_gcp_amd import runtime as rt, packing as pk
from video_gcp_amd.model import GCPTreeModel

hp = V.config("c1")
m = GCPTreeModel(hp, device="cuda")
lib = m.lib
S = 64
def run(R, Cin, tag):
    x = torch.randn(R, S, S, Cin, device="cuda")
    w = torch.randn(16, Cin, 3, 3)
    wpk = pk.pack_conv3x3(w, 16).cuda()
    out = torch.empty(R, S, S, 16, device="cuda")
    zeros = torch.zeros(64, device="cuda")
    a = m._conv_args([(x.data_ptr(), Cin, 1, None, None, rt.ACT_NONE)], R, S, S, S, S, 16, 16, wpk, zeros, out)
    st = torch.cuda.current_stream()
    for _ in range(2):
        rt.check(lib.gcpx_conv3x3(C.byref(a), st.cuda_stream), tag)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        rt.check(lib.gcpx_conv3x3(C.byref(a), st.cuda_stream), tag)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us")
run(1280, 112, "112ch x 1280 frames")
run(1280 * 7, 16, "16ch x 8960 frames")
run(1280, 16, "16ch x 1280 frames")
run(1280, 32, "32ch x 1280 frames")
run(1280, 64, "64ch x 1280 frames")
