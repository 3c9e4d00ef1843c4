"""Decoder 3x3 weight-gradient kernels at the c2 shapes, exact f32 vs split-f16: python tools/time_wgrad_conv.py [name-substring]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import video_gcp_amd as V
from video_gcp_amd import runtime as rt
lib = rt.load_library()
dev = torch.device("cuda")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
#        name            Cout Cin  S   F
CASES = [("head",        100, 16, 64, 1280), ("additional", 16, 32, 64, 2032), ("pyramid-0", 16, 64, 64, 2032),
         ("pyramid-1", 32, 128, 32, 2032), ("pyramid-2", 64, 256, 16, 2032)]
st = torch.cuda.current_stream().cuda_stream
cus = lib.gcpx_conv_grid() // 2
for name, Cout, Cin, S, Fr in CASES:
    if flt not in name:
        continue
    N16 = (Cout + 15) // 16 * 16
    dy = torch.randn(Fr, S, S, N16, device=dev); u = torch.randn(Fr, S, S, Cin, device=dev)
    ych = Cin // 32 if (Cin % 32 == 0 and N16 != 112) else Cin // 16
    per_cu = 2 if N16 in (112, 64) else 3
    grid = max(1, min(cus * per_cu // ych, Fr * max(1, S * S // 64)))
    part = torch.empty(grid, N16, 9 * Cin, device=dev)
    flop = 2.0 * 9 * Cin * Cout * S * S * Fr
    res = {}
    for rep in range(2):
        for nm, fn in (("f32", lib.gcpx_wgrad_conv3x3), ("split", lib.gcpx_wgrad_conv3x3_split)):
            for _ in range(2):
                rt.check(fn(dy.data_ptr(), N16, u.data_ptr(), Fr, S, S, Cin, Cout, part.data_ptr(), grid, st), nm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                rt.check(fn(dy.data_ptr(), N16, u.data_ptr(), Fr, S, S, Cin, Cout, part.data_ptr(), grid, st), nm)
            e1.record(); torch.cuda.synchronize()
            res[nm] = e0.elapsed_time(e1) / 5
    print(f"{name:12s} Cout {Cout:3d} Cin {Cin:3d} {S}x{S} F {Fr}: f32 {res['f32']*1e3:7.1f} us ({flop/res['f32']/1e9:6.1f} TF)   split {res['split']*1e3:7.1f} us ({flop/res['split']/1e9:6.1f} TF)  grid {grid}x{ych}")
