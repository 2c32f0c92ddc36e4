"""Vendor-library calibration of the matrix-core rate on THIS box (context for roofline.frac, which is priced against the 2.5 PFLOP/s
data-sheet figure): hipBLASLt bf16 GEMMs through torch.matmul, MIOpen's 3x3 convolution through torch.nn.functional.conv2d, and this
repo's resident-weight convolution on the same tensor.  python scripts/calib_mfma.py"""
import os, sys, json
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")


def t_us(f, iters=20, warm=5):
    for _ in range(warm): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


out = {}
for (M, N, K) in ((8192, 8192, 8192), (16384, 4096, 4096), (921600, 64, 576), (921600, 64, 64)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us = t_us(lambda: torch.matmul(a, b, out=c))
    out[f"hipblaslt_bf16_gemm_{M}x{N}x{K}"] = {"us": round(us, 1), "tflops": round(2.0 * M * N * K / us * 1e-6, 1)}
    print(f"hipBLASLt bf16 GEMM {M}x{N}x{K}: {us:.1f} us, {2.0 * M * N * K / us * 1e-6:.0f} TFLOP/s", flush=True)
    del a, b, c

B, H, W, Cn = 16, 180, 320, 64
x = torch.randn(B, Cn, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cn, Cn, 3, 3, device=dev) / 24).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
bias = torch.zeros(Cn, device=dev, dtype=torch.bfloat16)
flops = 2.0 * 9 * Cn * Cn * B * H * W
if os.environ.get("MIOPEN", "1") == "1":
    torch.backends.cudnn.benchmark = True
    us = t_us(lambda: F.conv2d(x, w, bias, padding=1), iters=10, warm=3)
    out["miopen_conv3x3_64_64_16x180x320_bf16_nhwc"] = {"us": round(us, 1), "tflops": round(flops / us * 1e-6, 1)}
    print(f"MIOpen conv 3x3 64->64 on 16x180x320 bf16 NHWC: {us:.1f} us, {flops / us * 1e-6:.0f} TFLOP/s", flush=True)

from fcvsr_amd import hip
xs = x.permute(0, 2, 3, 1)                                          # (B, H, W, C) dense view
dst = torch.empty_like(xs)
wp = hip.pack_conv_weight_mfma(w.float(), torch.bfloat16)
b32 = torch.zeros(Cn, device=dev)
us = t_us(lambda: hip.conv2d_mfma([dict(srcs=[xs], dst=dst)], wp, 3, Cn, hip.BF16, bias=b32))
out["fcvsr_conv3x3_64_64_16x180x320_bf16"] = {"us": round(us, 1), "tflops": round(flops / us * 1e-6, 1), "kernel": hip.lib().fcvsr_last_conv_kernel().decode()}
print(f"fcvsr conv 3x3 64->64 on 16x180x320 bf16 (one level): {us:.1f} us, {flops / us * 1e-6:.0f} TFLOP/s", flush=True)
print(json.dumps(out))
