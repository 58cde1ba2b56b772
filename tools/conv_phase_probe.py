"""Where do a conv workgroup's cycles go?  Builds conv3x3_bf16.hip with -DOSVOS_CONV_PROF (s_memtime marks around every
phase of the K loop) into a scratch library, runs one layer and prints the per-wave averages.  GPU box only.
usage: python tools/conv_phase_probe.py [--tile 1] [--batch 12] [--layer conv3_2]"""
import argparse, ctypes as C, os, subprocess
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--tile", type=int, default=1)
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--layer", default="conv3_2")
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
ap.add_argument("--lib", default="", help="prebuilt probe library (else built here with hipcc)")
ap.add_argument("--defs", default="", help="extra -D flags for the probe build, comma separated")
args = ap.parse_args()
LAYERS = {"conv1_2": (480, 854, 64, 64), "conv2_2": (240, 427, 128, 128), "conv3_2": (120, 214, 256, 256),
          "conv4_2": (60, 107, 512, 512), "conv5_2": (30, 54, 512, 512)}
H, W, Cin, Cout = LAYERS[args.layer]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "osvos-pytorch_amd", "csrc")
out = args.lib or "/tmp/libconvprof.so"
if not args.lib:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DOSVOS_CONV_PROF"]
                          + ["-D" + d for d in args.defs.split(",") if d]
                          + ([os.path.join(src, "conv3x3_bf16.hip"), os.path.join(src, "conv3x3_bf16_dma.hip")] if args.dtype == "bf16"
                             else [os.path.join(src, "conv3x3_f32.hip")])
                          + ["-x", "hip", os.path.join(src, "errors.cpp"), "-o", out])
lib = C.CDLL(out)
vp = C.c_void_p
dev = "cuda"
N = args.batch
x = torch.randn(N, H, W, Cin, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
y = torch.empty(N, H, W, Cout, device=dev)
CinP, CoutP = (Cin + 31) // 32 * 32, (Cout + 31) // 32 * 32
FP32 = args.dtype == "fp32"
if FP32:
    CinP = (Cin + 7) // 8 * 8
    wpk = torch.empty(9 * CinP * CoutP, dtype=torch.float32, device=dev)
    assert lib.osvos_prof_pack_fwd_f32(vp(w.data_ptr()), vp(wpk.data_ptr()), Cout, Cin) == 0
else:
    wpk = torch.empty(9 * CinP * CoutP, dtype=torch.int16, device=dev)
    assert lib.osvos_prof_pack_fwd_bf16(vp(w.data_ptr()), vp(wpk.data_ptr()), Cout, Cin) == 0
prof = torch.zeros(10 * (1 << 19), dtype=torch.int64, device=dev)
(lib.osvos_debug_set_conv_prof_f32 if FP32 else lib.osvos_debug_set_conv_prof)(vp(prof.data_ptr()))
def run():
    rc = (lib.osvos_prof_conv3x3_f32 if FP32 else lib.osvos_prof_conv3x3_bf16mfma)(vp(x.data_ptr()), vp(wpk.data_ptr()), vp(y.data_ptr()), N, H, W, Cin, Cout, args.tile)
    assert rc == 0, rc
for _ in range(3):
    run()
torch.cuda.synchronize()
prof.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
p = prof.cpu().numpy().reshape(-1, 10)
p = p[p[:, 9] != 0]
names = ["prologue+load0", "barrier1", "vmcnt wait", "cvt+ds_write", "barrier2", "issue loads", "mfma phase", "epilogue"]
nch = CinP // 32
if FP32:     # fp32 kernel: one barrier per 8-channel chunk, LDS double buffered
    names = ["prologue+chunk0", "barrier", "vmcnt wait", "ds_write", "-", "issue loads", "mfma phase", "epilogue"]
    nch = CinP // 8
tot = (p[:, 9] - p[:, 8]).astype(np.float64)
print(f"{args.layer} batch {N} tile {args.tile} {os.path.basename(out)}: {ms:.3f} ms, {2.0*N*H*W*Cin*Cout*9/ms/1e9:.0f} TFLOP/s, waves recorded {len(p)}, chunks {nch}")
print(f"wave lifetime mean {tot.mean():.0f} cycles (s_memtime ticks), kernel span {(p[:,9].max()-p[:,8].min())} ticks")
for k, nme in enumerate(names):
    print(f"  {nme:16s} {p[:,k].mean():10.0f}  {100*p[:,k].mean()/tot.mean():5.1f} %   per chunk {p[:,k].mean()/nch:8.0f}")
