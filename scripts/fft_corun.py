"""Reproducer hunt for DESIGN.md "Multi-stream replays": N rfft2 calls on one graph branch while another branch runs a
candidate co-runner; every output is compared with the solo result.  Prints mismatching replays per co-runner."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
L = hip.lib()
B, H, W, n = 2, int(os.environ.get("H", "36")), int(os.environ.get("W", "68")), 64
N = int(os.environ.get("N", "40"))
Wf = W // 2 + 1
torch.manual_seed(0)
x = torch.randn(B, H, W, n, device="cuda").to(torch.bfloat16)
xo = torch.randn(B, H, W, n, device="cuda").to(torch.bfloat16)

def ffts(src, outs):
    st = hip.stream_ptr()
    v = hip.view(src)
    for o in outs:
        hip.check(L.fcvsr_rfft2(C.byref(v), B, H, W, n, o.data_ptr(), 2 * n, 0, n, st), "rfft2")

ref = torch.empty(B, H, Wf, 2 * n, device="cuda")
ffts(x, [ref]); torch.cuda.synchronize()

w = torch.randn(64, 64, 3, 3, device="cuda") / 24
wp = hip.pack_conv_weight_mfma(w, torch.bfloat16)
csrc = torch.randn(B, H, W, 64, device="cuda").to(torch.bfloat16)
cdst = torch.empty(B, H, W, 64, device="cuda", dtype=torch.bfloat16)
ew = torch.randn(B, H, W, 64, device="cuda")
big = torch.randn(16, 180, 320, 64, device="cuda").to(torch.bfloat16)
bigd = torch.empty_like(big)

def co_none(): pass
def co_fft():
    ffts(xo, co_outs)
def co_conv():
    for _ in range(N):
        hip.conv2d_mfma([dict(srcs=[csrc], dst=cdst)], wp, 3, 64, hip.BF16, act=hip.ACT_LEAKY, slope=0.1)
def co_convbig():
    for _ in range(3):
        hip.conv2d_mfma([dict(srcs=[big], dst=bigd)], wp, 3, 64, hip.BF16, act=hip.ACT_LEAKY, slope=0.1)
bigf = torch.randn(8, 180, 320, 64, device="cuda")
bigfd = torch.empty_like(bigf)
w1 = torch.randn(64, 64, 1, 1, device="cuda") / 8
wp1 = hip.pack_conv_weight_mfma(w1, torch.bfloat16)
wd = hip.pack_conv_weight(w)
def co_conv_f32io():           # lean 3x3 kernel, f32 source and destination
    for _ in range(3):
        hip.conv2d_mfma([dict(srcs=[bigf], dst=bigfd)], wp, 3, 64, hip.BF16, act=hip.ACT_LEAKY, slope=0.1)
def co_conv1x1():              # lean 1x1 kernel
    for _ in range(6):
        hip.conv2d_mfma([dict(srcs=[big], dst=bigd)], wp1, 1, 64, hip.BF16)
def co_direct():               # direct f32 convolution: no MFMA
    hip.conv2d([bigf[:2]], wd, 3, 64, bigfd[:2])
def co_matmul():               # rocBLAS / hipBLASLt bf16 GEMM (MFMA, not this library's code)
    for _ in range(4):
        torch.matmul(mm_a, mm_b, out=mm_c)
mm_a = torch.randn(4096, 4096, device="cuda").to(torch.bfloat16); mm_b = torch.randn(4096, 4096, device="cuda").to(torch.bfloat16)
mm_c = torch.empty(4096, 4096, device="cuda", dtype=torch.bfloat16)
def co_eltwise():
    for _ in range(N):
        ew.mul_(1.0001)
co_outs = [torch.empty_like(ref) for _ in range(N)]
cands = dict(none=co_none, fft=co_fft, conv=co_conv, convbig=co_convbig, eltwise=co_eltwise, conv_f32io=co_conv_f32io,
             conv1x1=co_conv1x1, direct=co_direct, matmul=co_matmul)
only = os.environ.get("ONLY")
for name, co in cands.items():
    if only and name not in only.split(","):
        continue
    outs = [torch.empty_like(ref) for _ in range(N)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def run():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa): ffts(x, outs)
        with torch.cuda.stream(sb): co()
        cur.wait_stream(sa); cur.wait_stream(sb)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    bad_replays, bad_outs = 0, 0
    for r in range(int(os.environ.get("REPLAYS", "30"))):
        for o in outs: o.zero_()
        g.replay(); torch.cuda.synchronize()
        nb = sum(int(not torch.equal(o, ref)) for o in outs)
        bad_replays += nb > 0; bad_outs += nb
    print(f"co-runner {name:8s}: replays with a wrong rfft2 output {bad_replays}, wrong outputs {bad_outs} of {N * int(os.environ.get('REPLAYS', '30'))}", flush=True)
