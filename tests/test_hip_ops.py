"""GPU: individual C-ABI entry points against plain PyTorch fp32 references of the same op."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t):  # NCHW cpu -> NHWC cuda contiguous
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def _rand(*s, seed=0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(s).astype(np.float32))


@pytest.mark.parametrize("cin,cout,k,stride", [(7, 448, 3, 1), (64, 64, 3, 2), (4, 4, 7, 1), (64, 1, 3, 1), (84, 64, 3, 1)])
def test_conv_direct_vs_torch(cin, cout, k, stride):
    from fcvsr_amd import hip
    x, w, b = _rand(2, cin, 20, 28), _rand(cout, cin, k, k, seed=1) * 0.1, _rand(cout, seed=2)
    ref = F.leaky_relu(F.conv2d(x, w, b, stride=stride, padding=k // 2), 0.1)
    Ho, Wo = ref.shape[2:]
    dst = torch.empty(2, Ho, Wo, cout, device="cuda")
    src = x.cuda().permute(0, 2, 3, 1) if cin == 7 else nhwc(x)       # cin=7: strided NCHW view, like feat_extract
    hip.conv2d([src], hip.pack_conv_weight(w.cuda()), k, cout, dst, bias=b.cuda(), stride=stride, act=hip.ACT_LEAKY,
               slope=0.1)
    assert float((nchw(dst) - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("cins,cout,k,ps", [([64], 128, 3, False), ([128], 64, 3, False), ([64, 16, 4], 64, 3, False),
                                            ([64], 256, 1, True), ([64], 256, 3, True), ([128, 84], 64, 1, False),
                                            ([64], 1, 3, False), ([64], 576, 1, False)])
def test_conv_mfma_vs_torch(dt, cins, cout, k, ps):
    """Reference = fp32 conv of the operands rounded to the MFMA dtype (exactly what the kernel multiplies)."""
    from fcvsr_amd import hip
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    B, H, W = 2, 20, 44            # partial tiles in both directions; for 1x1 a flat tail (2*20*44 % 256 != 0)
    xs = [_rand(B, c, H, W, seed=10 + i) for i, c in enumerate(cins)]
    cin = sum(cins)
    w, b = _rand(cout, cin, k, k, seed=3) / np.sqrt(cin * k * k), _rand(cout, seed=4)
    res = _rand(B, cout, H, W, seed=5)
    xr = torch.cat(xs, 1).to(tdt).float()
    ref = F.conv2d(xr, w.to(tdt).float(), b, padding=k // 2)
    ref = torch.where(ref >= 0, ref, 0.2 * ref) + 0.5 * res
    if ps:
        ref = F.pixel_shuffle(ref, 2)
        dst = torch.empty(B, 2 * H, 2 * W, cout // 4, device="cuda")
    elif cout == 1:
        dst = torch.empty(B, cout, H, W, device="cuda").permute(0, 2, 3, 1)     # NCHW boundary view like conv_last0
    else:
        dst = torch.empty(B, H, W, cout, device="cuda")
    g = dict(srcs=[nhwc(x) for x in xs], dst=dst, res=[nhwc(res)])
    assert hip.mfma_eligible(k, 1, [g])
    bd = b.cuda()
    if ps:      # pixel-shuffled layers use sub-pixel-major row order for weights and bias (residuals stay natural)
        bd = bd[hip.ps_order(cout).cuda()].contiguous()
    hip.conv2d_mfma([g], hip.pack_conv_weight_mfma(w.cuda(), tdt, ps=ps), k, cout, hip.BF16 if dt == "bf16" else hip.F16,
                    bias=bd, act=hip.ACT_LEAKY, slope=0.2, res_scale=[0.5], pixel_shuffle=ps)
    err = float((nchw(dst) - ref).abs().max())
    assert err < 2e-5 * max(1.0, float(ref.abs().max())), err


def test_conv_mfma_stride2_and_planar_source():
    """rconcat-style stride-2 3x3 (evaluated at full resolution, even pixels kept) and the feat_extract case
    (7 NCHW frame planes read in place)."""
    from fcvsr_amd import hip
    tdt = torch.bfloat16
    x, w, b = _rand(2, 64, 22, 38), _rand(64, 64, 3, 3, seed=1) / 24, _rand(64, seed=2)
    ref = F.conv2d(x.to(tdt).float(), w.to(tdt).float(), b, stride=2, padding=1)
    dst = torch.empty(2, ref.shape[2], ref.shape[3], 64, device="cuda")
    g = dict(srcs=[nhwc(x)], dst=dst)
    assert hip.mfma_eligible(3, 2, [g])
    hip.conv2d_mfma([g], hip.pack_conv_weight_mfma(w.cuda(), tdt), 3, 64, hip.BF16, stride=2, bias=b.cuda())
    assert float((nchw(dst) - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    # planar NCHW source with 7 channels -> 448
    x, w, b = torch.rand(2, 7, 20, 44), _rand(448, 7, 3, 3, seed=3) / 8, _rand(448, seed=4)
    ref = F.conv2d(x.to(tdt).float(), w.to(tdt).float(), b, padding=1)
    xd = x.cuda()
    dst = torch.empty(2, 20, 44, 448, device="cuda")
    g = dict(srcs=[xd.permute(0, 2, 3, 1)], dst=dst)
    assert hip.mfma_eligible(3, 1, [g])
    hip.conv2d_mfma([g], hip.pack_conv_weight_mfma(w.cuda(), tdt), 3, 448, hip.BF16, bias=b.cuda())
    assert float((nchw(dst) - ref).abs().max()) < 2e-5 * float(ref.abs().max())


def test_conv_mfma_fused_context_block_partials():
    """ContextBlock pooling (softmax over H*W of <r, wmask>, :657-701) from the conv epilogue's per-wave partials ==
    the stand-alone two-stage kernel == the torch formula, all evaluated on the conv's own output r."""
    from fcvsr_amd import hip
    L = hip.lib()
    B, Cc, H, W = 2, 64, 22, 70                        # partial tiles: 22 = 5*4+2 rows, 70 = 2*32+6 cols
    x = nhwc(_rand(B, Cc, H, W))
    w = _rand(Cc, Cc, 3, 3, seed=1) / 24
    wm, w1, w2 = _rand(Cc, seed=2).cuda(), (_rand(Cc, Cc, seed=3) / 8).cuda(), (_rand(Cc, Cc, seed=4) / 8).cuda()
    r = torch.empty(B, H, W, Cc, device="cuda")
    nparts = ((H + 3) // 4) * ((W + 31) // 32)
    parts = torch.full((B, nparts, Cc + 2), float("nan"), device="cuda")
    hip.conv2d_mfma([dict(srcs=[x], dst=r, gc_partial=parts)], hip.pack_conv_weight_mfma(w.cuda(), torch.bfloat16), 3, Cc,
                    hip.BF16, gc_wmask=wm)
    add = torch.empty(B, Cc, device="cuda")
    hip.check(L.fcvsr_gc_finish(parts.data_ptr(), nparts, w1.data_ptr(), w2.data_ptr(), B, Cc, add.data_ptr(),
                                hip.stream_ptr()), "gc_finish")
    add2 = torch.empty(B, Cc, device="cuda")
    nblk = (H * W + 255) // 256
    scratch = torch.empty(B * nblk * (Cc + 2), device="cuda")
    hip.check(L.fcvsr_gc_context(r.data_ptr(), wm.data_ptr(), w1.data_ptr(), w2.data_ptr(), B, H, W, Cc, add2.data_ptr(),
                                 scratch.data_ptr(), scratch.numel(), hip.stream_ptr()), "gc_context")
    rr = r.cpu().reshape(B, H * W, Cc)
    m = torch.softmax(rr @ wm.cpu(), dim=1)
    ctx = (rr * m[..., None]).sum(1)
    ref = F.leaky_relu(ctx @ w1.cpu().T, 0.2) @ w2.cpu().T
    assert float((add.cpu() - ref).abs().max()) < 1e-5
    assert float((add2.cpu() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_conv_mfma_16bit_storage_is_bit_identical(dt):
    """A 16-bit intermediate between two MFMA convs gives exactly the result of an f32 intermediate (the consumer rounds
    its input to the MFMA dtype anyway)."""
    from fcvsr_amd import hip
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    mdt = hip.BF16 if dt == "bf16" else hip.F16
    x = nhwc(_rand(2, 64, 24, 40))
    w1 = hip.pack_conv_weight_mfma((_rand(128, 64, 3, 3, seed=1) / 24).cuda(), tdt)
    w2 = hip.pack_conv_weight_mfma((_rand(64, 128, 3, 3, seed=2) / 34).cuda(), tdt)
    outs = []
    for mid_dt in (torch.float32, tdt):
        mid = torch.empty(2, 24, 40, 128, device="cuda", dtype=mid_dt)
        out = torch.empty(2, 24, 40, 64, device="cuda")
        hip.conv2d_mfma([dict(srcs=[x], dst=mid)], w1, 3, 128, mdt, act=hip.ACT_LEAKY, slope=0.1)
        hip.conv2d_mfma([dict(srcs=[mid], dst=out)], w2, 3, 64, mdt)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


def test_conv_mfma_grouped_levels():
    from fcvsr_amd import hip
    w = _rand(64, 64, 3, 3, seed=1) / 24
    wp = hip.pack_conv_weight_mfma(w.cuda(), torch.bfloat16)
    xs = [_rand(1, 64, h, ww, seed=h) for h, ww in [(36, 64), (18, 32), (9, 16)]]
    groups = [dict(srcs=[nhwc(x)], dst=torch.empty(1, x.shape[2], x.shape[3], 64, device="cuda")) for x in xs]
    hip.conv2d_mfma(groups, wp, 3, 64, hip.BF16)
    for x, g in zip(xs, groups):
        ref = F.conv2d(x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), padding=1)
        assert float((nchw(g["dst"]) - ref).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("H,W,n", [(180, 320, 64), (20, 24, 64), (144, 176, 16), (272, 480, 8), (17, 23, 5), (64, 64, 12), (72, 81, 4),
                                   (81, 512, 4)])
def test_rfft2_irfft2_vs_torch(H, W, n):
    """Any length incl. primes 11/17/23 (Vid4 / CVCP sizes) and odd W; [imag, real] packing; c2r semantics."""
    from fcvsr_amd import hip
    L = hip.lib()
    x = _rand(1, n, H, W)
    Wf = W // 2 + 1
    src = nhwc(x)
    spec = torch.zeros(1, H, Wf, 2 * n, device="cuda")
    v = hip.view(src)
    hip.check(L.fcvsr_rfft2(C.byref(v), 1, H, W, n, spec.data_ptr(), 2 * n, 0, n, hip.stream_ptr()), "rfft2")
    X = torch.fft.rfft2(x.double())
    ref = torch.cat([X.imag, X.real], 1).float()
    scale = float(ref.abs().max())
    assert float((nchw(spec) - ref).abs().max()) < 3e-6 * scale
    # inverse of a NON-Hermitian-consistent spectrum (imag parts at DC/Nyquist must be ignored like torch's c2r)
    sp = _rand(1, 2 * n, H, Wf, seed=9)
    refi = torch.fft.irfft2(torch.complex(sp[:, n:].double(), sp[:, :n].double()), s=(H, W)).float()
    spec2 = nhwc(sp)
    dst = torch.empty(1, H, W, n, device="cuda")
    dv = hip.view(dst)
    hip.check(L.fcvsr_irfft2(spec2.data_ptr(), 2 * n, 0, n, 1, H, W, n, None, None, C.byref(dv), hip.stream_ptr()),
              "irfft2")
    assert float((nchw(dst) - refi).abs().max()) < 3e-6 * max(1.0, float(refi.abs().max()))


def test_fft_roundtrip_full_size():
    """Size-independent property at the benchmark size: irfft2(rfft2(x)) == x."""
    from fcvsr_amd import hip
    L = hip.lib()
    B, H, W, n = 2, 180, 320, 64
    src = torch.rand(B, H, W, n, device="cuda")
    spec = torch.empty(B, H, W // 2 + 1, 2 * n, device="cuda")
    dst = torch.empty_like(src)
    v, dv = hip.view(src), hip.view(dst)
    hip.check(L.fcvsr_rfft2(C.byref(v), B, H, W, n, spec.data_ptr(), 2 * n, 0, n, hip.stream_ptr()), "rfft2")
    hip.check(L.fcvsr_irfft2(spec.data_ptr(), 2 * n, 0, n, B, H, W, n, None, None, C.byref(dv), hip.stream_ptr()), "irfft2")
    assert float((dst - src).abs().max()) < 2e-6


def test_corr_warp_sac_vs_oracle():
    from fcvsr_amd import hip
    from oracle import fcvsr_oracle as O
    L = hip.lib()
    st = hip.stream_ptr()
    B, C2, H, Wf = 2, 128, 72, 19          # H > 68 exercises the row cut-off of the 64-row correlation image
    a, b = _rand(B, C2, H, Wf, seed=1), _rand(B, C2, H, Wf, seed=2)
    dst = torch.empty(B, H, Wf, 84, device="cuda")
    dv = hip.view(dst)
    ad, bd = nhwc(a), nhwc(b)      # keep the device tensors alive while the kernel runs
    hip.check(L.fcvsr_corr_lookup(ad.data_ptr(), bd.data_ptr(), C2, B, H, Wf, C2, 4, Wf, C.byref(dv), st), "corr")
    got = nchw(dst)
    assert torch.equal(got[:, 81:], torch.zeros_like(got[:, 81:]))
    ref = O.corr_lookup(a, b)
    assert float((got[:, :81] - ref).abs().max()) < 1e-6
    # the lookup vanishes beyond column radius+1 (the property the engine's strip evaluation rests on) ...
    assert torch.equal(ref[..., 6:], torch.zeros_like(ref[..., 6:]))
    # ... and a strip call (x_count < Wf) reproduces the leading columns of the full map
    strip = torch.empty(B, H, 8, 84, device="cuda")
    sv = hip.view(strip)
    hip.check(L.fcvsr_corr_lookup(ad.data_ptr(), bd.data_ptr(), C2, B, H, Wf, C2, 4, 8, C.byref(sv), st), "corr strip")
    assert torch.equal(strip, dst[:, :, :8])
    # warp + SAC
    B, Cc, H, W = 1, 64, 24, 40
    f, off, k1 = _rand(B, Cc, H, W, seed=3), _rand(B, 2, H, W, seed=4) * 3.0, _rand(B, 3 * Cc, H, W, seed=5)
    fd, od, kd = nhwc(f), nhwc(off), nhwc(k1)
    s = torch.empty_like(fd); vv = torch.empty_like(fd); out = torch.empty_like(fd)
    fv, ov, kv, sv, vvv, outv = (hip.view(t) for t in (fd, od, kd, s, vv, out))
    hip.check(L.fcvsr_warp(C.byref(fv), C.byref(ov), B, H, W, C.byref(sv), st), "warp")
    hip.check(L.fcvsr_sac_v(C.byref(sv), C.byref(kv), B, H, W, C.byref(vvv), st), "sac_v")
    hip.check(L.fcvsr_sac_h(C.byref(vvv), C.byref(kv), C.byref(fv), 0.1, B, H, W, C.byref(outv), st), "sac_h")
    sref = O.warp_bilinear(f, off)
    assert float((nchw(s) - sref).abs().max()) < 1e-5
    ref = F.leaky_relu(O.sac_kernel1_twice(sref, k1) + f, 0.1)
    assert float((nchw(out) - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("kdt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("Cc", [64, 32])             # 64: the 8-channels-per-lane kernel; 32: the 4-channel one
def test_iac_step_fused_vs_oracle(kdt, Cc):
    from fcvsr_amd import hip
    from oracle import fcvsr_oracle as O
    L = hip.lib()
    B, H, W = 2, 22, 37                              # partial tiles in both directions
    f, fin = _rand(B, Cc, H, W, seed=3), _rand(B, Cc, H, W, seed=6)
    off, k1 = _rand(B, 2, H, W, seed=4) * 3.0, _rand(B, 3 * Cc, H, W, seed=5)
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[kdt]
    k1r = k1.to(tdt).float()                         # the kernel sees K rounded to its storage dtype
    fd, find, od, kd = nhwc(f), nhwc(fin), nhwc(off), nhwc(k1).to(tdt)
    out = torch.empty_like(fd)
    fv, finv, ov, kv, outv = (hip.view(t) for t in (fd, find, od, kd, out))
    hip.check(L.fcvsr_iac_step(C.byref(fv), C.byref(ov), C.byref(kv), C.byref(finv), 0.1, B, H, W, C.byref(outv),
                               hip.stream_ptr()), "iac_step")
    ref = F.leaky_relu(O.sac_kernel1_twice(O.warp_bilinear(f, off), k1r) + fin, 0.1)
    assert float((nchw(out) - ref).abs().max()) < 2e-4


@pytest.mark.parametrize("adt", ["bf16", "f16"])
@pytest.mark.parametrize("Cc", [64, 32])
def test_iac_step_16bit_features(adt, Cc):
    """Features, adaptive kernels and the output stored in the 16-bit activation dtype (arithmetic stays f32): equal to
    the oracle on the rounded inputs up to the final rounding of the output."""
    from fcvsr_amd import hip
    from oracle import fcvsr_oracle as O
    L = hip.lib()
    B, H, W = 2, 21, 35
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[adt]
    f, fin = _rand(B, Cc, H, W, seed=13).to(tdt), _rand(B, Cc, H, W, seed=16).to(tdt)
    off, k1 = _rand(B, 2, H, W, seed=14) * 3.0, _rand(B, 3 * Cc, H, W, seed=15).to(tdt)
    fd, find, od, kd = nhwc(f.float()).to(tdt), nhwc(fin.float()).to(tdt), nhwc(off), nhwc(k1.float()).to(tdt)
    out = torch.empty_like(fd)
    fv, finv, ov, kv, outv = (hip.view(t) for t in (fd, find, od, kd, out))
    hip.check(L.fcvsr_iac_step(C.byref(fv), C.byref(ov), C.byref(kv), C.byref(finv), 0.1, B, H, W, C.byref(outv),
                               hip.stream_ptr()), "iac_step")
    ref = F.leaky_relu(O.sac_kernel1_twice(O.warp_bilinear(f.float(), off), k1.float()) + fin.float(), 0.1)
    got = nchw(out.float())
    eps = 2.0 ** -8 if adt == "bf16" else 2.0 ** -11
    assert float(((got - ref).abs() / (ref.abs() + 1.0)).max()) < 1.01 * eps


def test_flow_warp_known_answer():
    """The reference test-suite's own known answer for flow_warp (mmedit_train/tests/test_models/test_common/
    test_flow_warp.py:32-46): flow = -1 everywhere == shift by one pixel with zero fill."""
    from fcvsr_amd import hip
    L = hip.lib()
    x = torch.rand(1, 4, 10, 10)
    off = -torch.ones(1, 2, 10, 10)
    xd, od = nhwc(x), nhwc(off)
    out = torch.empty_like(xd)
    xv, ov, outv = hip.view(xd), hip.view(od), hip.view(out)
    hip.check(L.fcvsr_warp(C.byref(xv), C.byref(ov), 1, 10, 10, C.byref(outv), hip.stream_ptr()), "warp")
    ref = torch.zeros_like(x)
    ref[:, :, 1:, 1:] = x[:, :, :-1, :-1]
    assert float((nchw(out) - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("cin,cout,d16,nres,levels", [(64, 64, True, 0, [(21, 37)]), (64, 64, False, 2, [(21, 37), (11, 19), (6, 10)]),
                                                      (64, 128, True, 1, [(20, 70)]), (64, 256, True, 0, [(9, 33)]),
                                                      (128, 64, True, 1, [(21, 37), (11, 19)]), (128, 64, False, 0, [(16, 64)]),
                                                      (64, 64, True, 2, [(180, 320), (90, 160), (45, 80)])])
def test_conv_resident_weights_matches_lean(cin, cout, d16, nres, levels, monkeypatch):
    """The LDS-resident-weight 3x3 kernel (conv_res.hip; FCVSR_MFMA_RES=1 forces it at any size) accumulates the 16-deep
    k-steps in the same order as the lean kernel, so the two must agree bit for bit (partial tiles, three grouped levels,
    residuals, both destination types, both input-channel counts, odd tile counts per workgroup)."""
    from fcvsr_amd import hip
    dt = torch.bfloat16
    g0 = torch.Generator().manual_seed(cout + nres + cin)
    w = torch.randn(cout, cin, 3, 3, generator=g0) / (3.0 * cin ** 0.5)
    bias = torch.randn(cout, generator=g0).cuda()
    wp = hip.pack_conv_weight_mfma(w.cuda(), dt)
    groups = []
    B = 2 if levels[0][0] < 100 else 3
    for (H, W) in levels:
        x = torch.randn(B, H, W, cin, generator=g0).cuda().to(dt)
        y = torch.empty(B, H, W, cout, device="cuda", dtype=dt if d16 else torch.float32)
        res = [torch.randn(B, H, W, cout, generator=g0).cuda().to(dt) for _ in range(nres)]
        groups.append(dict(srcs=[x], dst=y, res=res))
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("FCVSR_MFMA_RES", mode)
        for g in groups:
            g["dst"].fill_(float("nan"))
        hip.conv2d_mfma(groups, wp, 3, cout, hip.BF16, bias=bias, act=hip.ACT_LEAKY, slope=0.1, res_scale=[1.0, -0.5][:nres])
        torch.cuda.synchronize()
        outs.append([g["dst"].clone() for g in groups])
    monkeypatch.delenv("FCVSR_MFMA_RES")
    for a, b in zip(*outs):
        assert not torch.isnan(a.float()).any() and not torch.isnan(b.float()).any()
        assert torch.equal(a, b)


@pytest.mark.parametrize("adt", ["f32", "bf16"])
def test_iac_step2_equals_two_single_steps(adt):
    """The two-direction launch (K1 read once) must reproduce two single-direction launches bit for bit."""
    from fcvsr_amd import hip
    L = hip.lib()
    B, Cc, H, W = 2, 64, 19, 41
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16}[adt]
    prev = [nhwc(_rand(B, Cc, H, W, seed=21 + d)).to(tdt) for d in range(2)]
    fin = [nhwc(_rand(B, Cc, H, W, seed=31 + d)).to(tdt) for d in range(2)]
    offs = nhwc(_rand(B, 4, H, W, seed=41) * 2.5)                  # channels (0,1): forward, (2,3): backward
    k1 = nhwc(_rand(B, 3 * Cc, H, W, seed=51)).to(tdt)
    kv = hip.view(k1)
    ref = []
    for d in range(2):
        out = torch.empty_like(prev[d])
        pv, ov, fv, dv = hip.view(prev[d]), hip.view(offs[..., 2 * d:2 * d + 2]), hip.view(fin[d]), hip.view(out)
        hip.check(L.fcvsr_iac_step(C.byref(pv), C.byref(ov), C.byref(kv), C.byref(fv), 0.1, B, H, W, C.byref(dv),
                                   hip.stream_ptr()), "iac_step")
        ref.append(out)
    V2 = hip.View * 2
    outs = [torch.empty_like(prev[0]), torch.empty_like(prev[1])]
    hip.check(L.fcvsr_iac_step2(V2(hip.view(prev[0]), hip.view(prev[1])),
                                V2(hip.view(offs[..., 0:2]), hip.view(offs[..., 2:4])), C.byref(kv),
                                V2(hip.view(fin[0]), hip.view(fin[1])), 0.1, B, H, W,
                                V2(hip.view(outs[0]), hip.view(outs[1])), hip.stream_ptr()), "iac_step2")
    torch.cuda.synchronize()
    assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1])


@pytest.mark.parametrize("adt,kdt,shape", [("bf16", "bf16", (2, 18, 37)), ("f32", "bf16", (2, 18, 37)), ("f16", "f16", (2, 18, 37)),
                                           ("bf16", "bf16", (1, 4, 14)), ("bf16", "bf16", (3, 5, 3)), ("bf16", "bf16", (1, 9, 29)),
                                           ("f16", "f16", (2, 7, 15)), ("bf16", "bf16", (1, 33, 71))])
def test_iac_step2_fused_predictor_equals_unfused(adt, kdt, shape):
    """F[1] folded into the IAC kernel: same result as the stand-alone 1x1 MFMA convolution (16-bit K) followed by the
    two-direction IAC launch - the fold changes where the kernels live (registers / LDS instead of HBM), not their values.
    16-bit activations take iac_fused2_kernel (kernels in MFMA accumulator layout, 4 x 14 tiles: the shapes cover one exact
    tile, partial tiles in both directions and images smaller than a tile), f32 activations iac_step64_kernel."""
    from fcvsr_amd import hip
    L = hip.lib()
    Cc = 64
    B, H, W = shape
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[adt]
    mdt = {"bf16": torch.bfloat16, "f16": torch.float16}[kdt]
    mcode = hip.BF16 if kdt == "bf16" else hip.F16
    prev = [nhwc(_rand(B, Cc, H, W, seed=61 + d)).to(tdt) for d in range(2)]
    fin = [nhwc(_rand(B, Cc, H, W, seed=71 + d)).to(tdt) for d in range(2)]
    offs = nhwc(_rand(B, 4, H, W, seed=81) * 2.5)
    k0 = nhwc(_rand(B, Cc, H, W, seed=91)).to(mdt)
    w = _rand(3 * 3 * Cc, Cc, 1, 1, seed=92).cuda() / 8.0         # 3 iterations x 192 rows
    bias = _rand(3 * 3 * Cc, seed=93).cuda() * 0.1
    wp = hip.pack_conv_weight_mfma(w, mdt)
    K = torch.empty(B, H, W, 3 * 3 * Cc, device="cuda", dtype=mdt)
    hip.conv2d_mfma([dict(srcs=[k0], dst=K)], wp, 1, 3 * 3 * Cc, mcode, bias=bias)
    V2 = hip.View * 2
    it = 1                                                          # middle iteration: non-zero row / bias offsets
    kv = hip.view(K[..., it * 192:(it + 1) * 192])
    ref = [torch.empty_like(prev[0]), torch.empty_like(prev[1])]
    args = (V2(hip.view(prev[0]), hip.view(prev[1])), V2(hip.view(offs[..., 0:2]), hip.view(offs[..., 2:4])))
    fins = V2(hip.view(fin[0]), hip.view(fin[1]))
    hip.check(L.fcvsr_iac_step2(args[0], args[1], C.byref(kv), fins, 0.1, B, H, W,
                                V2(hip.view(ref[0]), hip.view(ref[1])), hip.stream_ptr()), "iac_step2")
    out = [torch.empty_like(prev[0]), torch.empty_like(prev[1])]
    k0v = hip.view(k0)
    hip.check(L.fcvsr_iac_step2_fused(args[0], args[1], C.byref(k0v), wp.data_ptr() + it * 192 * 64 * 2,
                                      bias.data_ptr() + it * 192 * 4, fins, 0.1, B, H, W,
                                      V2(hip.view(out[0]), hip.view(out[1])), hip.stream_ptr()), "iac_step2_fused")
    torch.cuda.synchronize()
    for o, r_ in zip(out, ref):
        assert torch.equal(o, r_)


def test_freq_mlp3_matches_three_1x1_launches():
    """The fused convfuse stack against the three stand-alone MFMA 1x1 launches (bf16 operands, f32 accumulate, hidden
    tensors rounded to bf16 in both): identical accumulation order per layer, so the results must agree exactly."""
    from fcvsr_amd import hip
    L = hip.lib()
    B, H, Wf, n = 2, 9, 23, 64                              # 414 pixels: partial last workgroup
    spec = nhwc(_rand(B, 6 * n, H, Wf, seed=101))
    x1f, x2f, x3f = spec[..., :2 * n], spec[..., 2 * n:4 * n], spec[..., 4 * n:]
    dt = torch.bfloat16
    w0 = _rand(128, 256, 1, 1, seed=102).cuda() / 16
    w2 = _rand(128, 128, 1, 1, seed=103).cuda() / 11
    w4 = _rand(128, 128, 1, 1, seed=104).cuda() / 11
    p0, p2, p4 = (hip.pack_conv_weight_mfma(w, dt) for w in (w0, w2, w4))
    ref = torch.empty(2 * B, H, Wf, 2 * n, device="cuda", dtype=dt)
    t0 = torch.empty(2 * B, H, Wf, 2 * n, device="cuda", dtype=dt)
    t1 = torch.empty_like(t0)
    dirs = list(enumerate((x1f, x3f)))
    hip.conv2d_mfma([dict(srcs=[xa, x2f], dst=t0[d * B:(d + 1) * B]) for d, xa in dirs], p0, 1, 128, hip.BF16, act=hip.ACT_RELU)
    hip.conv2d_mfma([dict(srcs=[t0], dst=t1)], p2, 1, 128, hip.BF16, act=hip.ACT_RELU)
    hip.conv2d_mfma([dict(srcs=[t1[d * B:(d + 1) * B]], dst=ref[d * B:(d + 1) * B], res=[xa, x2f]) for d, xa in dirs], p4, 1, 128,
                    hip.BF16, res_scale=[1.0, -1.0])
    out = torch.zeros_like(ref)
    P2 = C.c_void_p * 2
    hip.check(L.fcvsr_freq_mlp3(P2(x1f.data_ptr(), x3f.data_ptr()), P2(x2f.data_ptr(), x2f.data_ptr()), 2, 6 * n, B * H * Wf,
                                p0.data_ptr(), p2.data_ptr(), p4.data_ptr(), P2(out[:B].data_ptr(), out[B:].data_ptr()), 2 * n,
                                hip.stream_ptr()), "freq_mlp3")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("k", [1, 3, 5, 7])
def test_convblk_fused_head_matches_separate_launches(k):
    """fcvsr_convblk (conv1 + PReLU + conv2 + channel sums | gate + tail) against the six stand-alone launches it replaces."""
    from fcvsr_amd import hip
    L = hip.lib()
    st = hip.stream_ptr()
    B, H, Wf, A, i = 2, 21, 19, 3, 1                         # partial 16x16 tiles in both directions
    x = nhwc(_rand(2 * B, 4, H, Wf, seed=200 + k))
    sim = nhwc(_rand(B, 4, H, Wf, seed=201))
    w1 = _rand(4, 4, k, k, seed=202).cuda() / (2.0 * k)
    w2 = _rand(4, 4, k, k, seed=203).cuda() / (2.0 * k)
    slope = torch.tensor([0.25], device="cuda")
    cw1, cw2 = _rand(4, 4, seed=204).cuda(), _rand(4, 4, seed=205).cuda()
    p1, p2 = hip.pack_conv_weight(w1), hip.pack_conv_weight(w2)
    # separate launches
    tt, u = torch.empty_like(x), torch.empty_like(x)
    hip.conv2d([x], p1, k, 4, tt, act=hip.ACT_PRELU, slope_t=slope)
    hip.conv2d([tt], p2, k, 4, u)
    nblk = (H * Wf + 255) // 256
    sums, scratch = torch.empty(2 * B, 4, device="cuda"), torch.empty(2 * B * nblk * 4, device="cuda")
    uv = hip.view(u)
    hip.check(L.fcvsr_channel_sum(C.byref(uv), 2 * B, H, Wf, sums.data_ptr(), scratch.data_ptr(), scratch.numel(), st), "sum")
    gate = torch.empty(2 * B, 4, device="cuda")
    hip.check(L.fcvsr_ca_gate(sums.data_ptr(), 1.0 / (H * Wf), cw1.data_ptr(), cw2.data_ptr(), 2 * B, 4, 4, gate.data_ptr(), st), "gate")
    ref = torch.zeros(B, H, Wf, 8 * A, device="cuda")
    hip.check(L.fcvsr_convblk_tail(u.data_ptr(), gate.data_ptr(), sim.data_ptr(), B, 2, H, Wf, ref.data_ptr(), 8 * A, 0, 4 * A, A, i,
                                   st), "tail")
    # fused
    out = torch.zeros_like(ref)
    u2 = torch.empty_like(x)
    ntile = ((H + 15) // 16) * ((Wf + 15) // 16)
    part = torch.empty(2 * B * ntile * 4, device="cuda")
    hip.check(L.fcvsr_convblk(x.data_ptr(), p1.data_ptr(), p2.data_ptr(), slope.data_ptr(), k, cw1.data_ptr(), cw2.data_ptr(),
                              sim.data_ptr(), B, 2, H, Wf, u2.data_ptr(), part.data_ptr(), part.numel(), out.data_ptr(), 8 * A, 0,
                              4 * A, A, i, st), "convblk")
    torch.cuda.synchronize()
    assert float((u2 - u).abs().max()) < 1e-5
    assert float((out - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("nhid,src", [(2, "bf16"), (1, "f32")])
def test_freq_head_matches_separate_1x1_launches(nhid, src):
    """fcvsr_freq_head (convcorr: 128 -> 64 -> 64 -> 4 on bf16 input; convcrt: 128 -> 64 -> 4 on an f32 spectrum slice)
    against the stand-alone MFMA 1x1 launches with bf16 hidden tensors."""
    from fcvsr_amd import hip
    L = hip.lib()
    npix_shape = (2, 9, 23)                                 # 414 pixels: partial last workgroup
    dt = torch.bfloat16
    if src == "bf16":
        x = nhwc(_rand(npix_shape[0], 128, *npix_shape[1:], seed=301)).to(dt)
        xs, sx, code = x, 128, hip.BF16
    else:
        spec = nhwc(_rand(npix_shape[0], 384, *npix_shape[1:], seed=302))
        xs, sx, code = spec[..., 128:256], 384, hip.F32
    w0 = _rand(64, 128, 1, 1, seed=303).cuda() / 11
    w1 = _rand(64, 64, 1, 1, seed=304).cuda() / 8
    wl = _rand(4, 64, 1, 1, seed=305).cuda() / 8
    p0, p1, pl = (hip.pack_conv_weight_mfma(w, dt) for w in (w0, w1, wl))
    B, H, Wf = npix_shape
    t0 = torch.empty(B, H, Wf, 64, device="cuda", dtype=dt)
    t1 = torch.empty_like(t0)
    ref = torch.empty(B, H, Wf, 4, device="cuda")
    hip.conv2d_mfma([dict(srcs=[xs], dst=t0)], p0, 1, 64, hip.BF16, act=hip.ACT_RELU)
    last = t0
    if nhid == 2:
        hip.conv2d_mfma([dict(srcs=[t0], dst=t1)], p1, 1, 64, hip.BF16, act=hip.ACT_RELU)
        last = t1
    hip.conv2d_mfma([dict(srcs=[last], dst=ref)], pl, 1, 4, hip.BF16)
    out = torch.zeros_like(ref)
    hip.check(L.fcvsr_freq_head(xs.data_ptr(), code, sx, B * H * Wf, p0.data_ptr(), p1.data_ptr() if nhid == 2 else None,
                                pl.data_ptr(), out.data_ptr(), hip.stream_ptr()), "freq_head")
    torch.cuda.synchronize()
    assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.parametrize("cin,cout,k,H,W,nres,act", [(64, 64, 3, 21, 37, 0, "leaky"), (128, 64, 3, 18, 70, 2, "none"), (64, 1152, 1, 13, 29, 0, "none"),
                                                    (32, 36, 1, 9, 17, 1, "relu"), (64, 128, 3, 40, 33, 1, "prelu")])
def test_conv_f32_matrix_core_kernel_equals_direct_kernel(cin, cout, k, H, W, nres, act):
    """fcvsr_conv2d_f32mfma (v_mfma_f32_32x32x2_f32, exact f32 products and sums) against the direct VALU kernel on the same
    layer: only the summation order differs (tolerance 2e-6 of the output scale), incl. partial tiles, bias, every activation
    and two scaled residuals."""
    from fcvsr_amd import hip
    g0 = torch.Generator().manual_seed(cin + cout + k)
    w = (torch.randn(cout, cin, k, k, generator=g0) / (cin * k * k) ** 0.5).cuda()
    bias = torch.randn(cout, generator=g0).cuda()
    x = torch.randn(2, H, W, cin, generator=g0).cuda()
    res = [torch.randn(2, H, W, cout, generator=g0).cuda() for _ in range(nres)]
    slope_t = torch.tensor([0.3]).cuda()
    a = {"leaky": hip.ACT_LEAKY, "none": hip.ACT_NONE, "relu": hip.ACT_RELU, "prelu": hip.ACT_PRELU}[act]
    kw = dict(bias=bias, act=a, slope=0.1, slope_t=slope_t if act == "prelu" else None, res=res, res_scale=[1.0, -0.5][:nres])
    wd, wm = hip.pack_conv_weight(w), hip.pack_conv_weight_f32mfma(w)
    y_d = hip.conv2d([x], wd, k, cout, torch.empty(2, H, W, cout, device="cuda"), **kw).clone()
    y_m = torch.full((2, H, W, cout), float("nan"), device="cuda")
    d = hip.ConvDesc()
    hip._fill_desc(d, [x], wm, k, cout, wm.shape[1], y_m, bias, 1, a, 0.1, kw["slope_t"], res, kw["res_scale"], False)
    import ctypes as C
    assert hip.lib().fcvsr_conv2d_f32mfma_eligible(C.byref(d)) == 1
    hip.conv2d([x], wd, k, cout, y_m, w_f32mfma=wm, **kw)
    torch.cuda.synchronize()
    assert not torch.isnan(y_m).any()
    assert float((y_m - y_d).abs().max()) <= 2e-6 * max(1.0, float(y_d.abs().max()))


def test_conv_f32_matrix_core_pixel_shuffle_and_skinny_outputs():
    """The two remaining heavy layers of the exact-f32 up-sampler on the f32-operand matrix-core kernel: a 1x1 64->256 layer with
    PReLU and PixelShuffle(2) store (sub-pixel-major packed rows) and the 3x3 64->C_img layer that adds into the NCHW result
    (scalar epilogue, strided destination, residual = destination), both against the direct kernel."""
    from fcvsr_amd import hip
    g0 = torch.Generator().manual_seed(9)
    B, H, W = 2, 10, 37
    x = torch.randn(B, H, W, 64, generator=g0).cuda()
    slope_t = torch.tensor([0.25]).cuda()
    for k in (1, 3):
        w = (torch.randn(256, 64, k, k, generator=g0) / (64 * k * k) ** 0.5).cuda()
        bias = torch.randn(256, generator=g0).cuda()
        y_d = hip.conv2d([x], hip.pack_conv_weight(w), k, 256, torch.empty(B, 2 * H, 2 * W, 64, device="cuda"), bias=bias,
                         act=hip.ACT_PRELU, slope_t=slope_t, pixel_shuffle=True).clone()
        y_m = torch.full((B, 2 * H, 2 * W, 64), float("nan"), device="cuda")
        hip.conv2d([x], hip.pack_conv_weight(w), k, 256, y_m, bias=bias, act=hip.ACT_PRELU, slope_t=slope_t, pixel_shuffle=True,
                   w_f32mfma=hip.pack_conv_weight_f32mfma(w, ps=True), bias_f32mfma=bias[hip.ps_order(256).cuda()].contiguous())
        torch.cuda.synchronize()
        assert not torch.isnan(y_m).any()
        assert float((y_m - y_d).abs().max()) <= 2e-6 * max(1.0, float(y_d.abs().max()))
    for cimg in (1, 3):
        w = (torch.randn(cimg, 64, 3, 3, generator=g0) / 24.0).cuda()
        bias = torch.randn(cimg, generator=g0).cuda()
        base = torch.randn(B, cimg, H, W, generator=g0).cuda()
        out_d, out_m = base.clone(), base.clone()
        for out, wm in ((out_d, None), (out_m, hip.pack_conv_weight_f32mfma(w))):
            ov = out.permute(0, 2, 3, 1)
            hip.conv2d([x], hip.pack_conv_weight(w), 3, cimg, ov, bias=bias, res=[ov], w_f32mfma=wm)
        torch.cuda.synchronize()
        assert float((out_m - out_d).abs().max()) <= 2e-6 * max(1.0, float(out_d.abs().max()))
        assert float((out_d - base).abs().max()) > 0.1


def test_gc_partials_from_stored_r_match_the_reference_formula():
    """fcvsr_gc_partial_levels: per 4 x 32 tile, (sum_p exp(l_p - m) r_p, m, sum exp) of a stored bf16 r over three levels with
    partial tiles; combined by fcvsr_gc_finish_levels they must give the ContextBlock vector of the f64 formula (:657-701)."""
    from fcvsr_amd import hip
    L = hip.lib()
    g0 = torch.Generator().manual_seed(3)
    B, n = 2, 64
    shapes = [(22, 70), (11, 35), (6, 18)]
    wmask = (torch.randn(n, generator=g0) * 0.3).cuda()
    w1 = (torch.randn(n, n, generator=g0) / 8).cuda()
    w2 = (torch.randn(n, n, generator=g0) / 8).cuda()
    rs = [torch.randn(B, H, W, n, generator=g0).cuda().to(torch.bfloat16) for H, W in shapes]
    nparts = [((H + 3) // 4) * ((W + 31) // 32) for H, W in shapes]
    parts = [torch.full((B, npt, n + 2), float("nan"), device="cuda") for npt in nparts]
    adds = [torch.empty(B, n, device="cuda") for _ in shapes]
    pl = (hip.GcPartialLevel * 3)()
    fl = (hip.GcFinishLevel * 3)()
    for l, (H, W) in enumerate(shapes):
        pl[l].r, pl[l].partial, pl[l].B, pl[l].H, pl[l].W = rs[l].data_ptr(), parts[l].data_ptr(), B, H, W
        fl[l].partial, fl[l].add, fl[l].nparts = parts[l].data_ptr(), adds[l].data_ptr(), nparts[l]
    hip.check(L.fcvsr_gc_partial_levels(pl, 3, hip.BF16, wmask.data_ptr(), n, hip.stream_ptr()), "gc_partial_levels")
    hip.check(L.fcvsr_gc_finish_levels(fl, 3, w1.data_ptr(), w2.data_ptr(), B, n, hip.stream_ptr()), "gc_finish_levels")
    torch.cuda.synchronize()
    for l, r in enumerate(rs):
        assert not torch.isnan(parts[l]).any()
        rd = r.double().reshape(B, -1, n)
        logits = rd @ wmask.double()
        ctx = (torch.softmax(logits, 1).unsqueeze(-1) * rd).sum(1)
        t = ctx @ w1.double().t()
        t = torch.where(t >= 0, t, 0.2 * t)
        ref = t @ w2.double().t()
        assert float((adds[l].double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_conv_resident_weights_pixel_shuffle_matches_generic(monkeypatch):
    """3x3 64 -> 256 with PReLU and PixelShuffle(2) (the up-convs of the full / RGB models) on the resident-weight kernel - a
    64-cout block is one sub-pixel - against the generic MFMA kernel's pixel-shuffle epilogue, bit for bit."""
    from fcvsr_amd import hip
    dt = torch.bfloat16
    g0 = torch.Generator().manual_seed(12)
    w = torch.randn(256, 64, 3, 3, generator=g0) / 24.0
    bias = torch.randn(256, generator=g0)
    wp = hip.pack_conv_weight_mfma(w.cuda(), dt, ps=True)
    bp = bias[hip.ps_order(256)].contiguous().cuda()
    slope_t = torch.tensor([0.25]).cuda()
    x = torch.randn(3, 37, 70, 64, generator=g0).cuda().to(dt)
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("FCVSR_MFMA_RES", mode)
        y = torch.full((3, 74, 140, 64), float("nan"), device="cuda").to(dt)
        hip.conv2d_mfma([dict(srcs=[x], dst=y, ps=True)], wp, 3, 256, hip.BF16, bias=bp, act=hip.ACT_PRELU, slope_t=slope_t,
                        pixel_shuffle=True)
        torch.cuda.synchronize()
        assert hip.lib().fcvsr_last_conv_kernel().decode().startswith("conv3_res" if mode == "1" else "conv_mfma")
        outs.append(y.clone())
    monkeypatch.delenv("FCVSR_MFMA_RES")
    assert not torch.isnan(outs[0].float()).any() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("cimg", [1, 3])
def test_conv_last_kernel_vs_f32_reference(cimg):
    """fcvsr_conv_last (3x3, 64 -> 1 or 3 channels, taps as MFMA columns, read-modify-write of the NCHW result) against torch's f32
    convolution of the same bf16-rounded operands: bf16 products are exact in f32, only the summation order differs."""
    import ctypes as C
    from fcvsr_amd import hip
    g0 = torch.Generator().manual_seed(40 + cimg)
    B, H, W = 2, 21, 45                                      # partial tiles in both directions
    u = torch.randn(B, H, W, 64, generator=g0).to(torch.bfloat16)
    w = (torch.randn(cimg, 64, 3, 3, generator=g0) / 24.0).to(torch.bfloat16)
    bias = torch.randn(cimg, generator=g0)
    base = torch.randn(B, cimg, H, W, generator=g0)
    ref = base + torch.nn.functional.conv2d(u.float().permute(0, 3, 1, 2), w.float(), bias, padding=1)
    tab = torch.zeros(16 if cimg == 1 else 32, 64)
    tab[:9 * cimg] = w.float().permute(2, 3, 0, 1).reshape(9 * cimg, 64)
    out = base.clone().cuda()
    ud, td, bd = u.cuda(), tab.to(torch.bfloat16).cuda(), bias.cuda()
    uv, ov = hip.view(ud), hip.view(out.permute(0, 2, 3, 1))
    hip.check(hip.lib().fcvsr_conv_last(C.byref(uv), td.data_ptr(), bd.data_ptr(), B, H, W, cimg, C.byref(ov), hip.stream_ptr()), "conv_last")
    torch.cuda.synchronize()
    assert float((out.cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("H,W", [(22, 70), (2, 2), (36, 64)])
def test_rcb_level0_equals_apply_then_xscale(dt, H, W):
    """fcvsr_rcb_level0 (R0 never stored) against the two-kernel sequence fcvsr_gc_apply_levels -> fcvsr_xscale_levels it
    replaces at the full-resolution level (reference BlockRCB :722-725, :766-777): bit-identical out and pooled R, including
    the clamped borders of the bilinear x2 up-sample; plus an f64 restatement of the formula as an independent check."""
    from fcvsr_amd import hip
    L = hip.lib()
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    code = hip.BF16 if dt == "bf16" else hip.F16
    g0 = torch.Generator().manual_seed(H * 100 + W)
    B, n = 2, 64
    mk = lambda h, w: torch.randn(B, h, w, n, generator=g0).cuda().to(tdt)
    x, r, z, up = mk(H, W), mk(H, W), mk(H, W), mk(H // 2, W // 2)
    add = torch.randn(B, n, generator=g0).cuda()
    st = hip.stream_ptr()
    # two-kernel sequence
    R = torch.empty_like(x); P_ref = torch.empty(B, H // 2, W // 2, n, device="cuda", dtype=tdt); out_ref = torch.empty_like(x)
    al = (hip.GcApplyLevel * 3)()
    al[0].r, al[0].add, al[0].z, al[0].out, al[0].pool = r.data_ptr(), add.data_ptr(), z.data_ptr(), R.data_ptr(), P_ref.data_ptr()
    al[0].B, al[0].H, al[0].W = B, H, W
    hip.check(L.fcvsr_gc_apply_levels(al, 1, code, code, 0.2, n, st), "gc_apply_levels")
    xl = (hip.XscaleLevel * 3)()
    xl[0].x, xl[0].r, xl[0].out, xl[0].dn, xl[0].up = x.data_ptr(), R.data_ptr(), out_ref.data_ptr(), None, up.data_ptr()
    xl[0].r_scale, xl[0].dn_pooled, xl[0].B, xl[0].H, xl[0].W = 2.0, 1, B, H, W
    hip.check(L.fcvsr_xscale_levels(xl, 1, code, n, st), "xscale_levels")
    # one pass
    P = torch.full_like(P_ref, float("nan")); out = torch.full_like(x, float("nan"))
    hip.check(L.fcvsr_rcb_level0(x.data_ptr(), r.data_ptr(), add.data_ptr(), z.data_ptr(), up.data_ptr(), out.data_ptr(),
                                 P.data_ptr(), 0.2, 2.0, code, B, H, W, n, st), "rcb_level0")
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16))
    assert torch.equal(P.view(torch.int16), P_ref.view(torch.int16))
    # independent f64 formula
    v = r.double() + add.double()[:, None, None, :]
    Rd = (torch.where(v >= 0, v, 0.2 * v).float() + z.float()).to(tdt).double()
    upd = F.interpolate(up.double().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    ref = x.double() + 2.0 * Rd + upd
    tol = (2.0 ** -8 if dt == "bf16" else 2.0 ** -11) * float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) <= tol
    pd = F.avg_pool2d(Rd.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert float((P.double() - pd).abs().max()) <= (2.0 ** -8 if dt == "bf16" else 2.0 ** -11) * float(pd.abs().max())


@pytest.mark.parametrize("H,W,n,Q", [(180, 320, 64, 4), (64, 72, 12, 3), (20, 24, 8, 2)])
def test_irfft2_bands_equals_band_by_band(H, W, n, Q):
    """fcvsr_irfft2_bands (spectrum columns read once for all masks when H has a two-stage factorisation; otherwise the plain
    loop) against Q calls of fcvsr_irfft2 with the same masks: bit-identical bands (reference Split_freq :2082-2090)."""
    from fcvsr_amd import hip
    L = hip.lib()
    g0 = torch.Generator().manual_seed(H + W + n)
    B, Wf = 2, W // 2 + 1
    src = torch.randn(B, H, W, n, generator=g0).cuda()
    spec = torch.empty(B, H, Wf, 2 * n, device="cuda")
    sv = hip.view(src)
    st = hip.stream_ptr()
    hip.check(L.fcvsr_rfft2(C.byref(sv), B, H, W, n, spec.data_ptr(), 2 * n, 0, n, st), "rfft2")
    masks = torch.rand(Q, H, Wf, generator=g0).cuda()
    ref = torch.empty(Q, B, H, W, n, device="cuda")
    work = torch.empty(B, H, Wf, 2 * n, device="cuda")
    for q in range(Q):
        bv = hip.view(ref[q])
        hip.check(L.fcvsr_irfft2(spec.data_ptr(), 2 * n, 0, n, B, H, W, n, masks[q].data_ptr(), work.data_ptr(), C.byref(bv), st), "irfft2")
    out = torch.full((Q, B, H, W, n), float("nan"), device="cuda")
    workq = torch.empty(Q, B, H, Wf, 2 * n, device="cuda")
    bvs = (hip.View * Q)(*[hip.view(out[q]) for q in range(Q)])
    hip.check(L.fcvsr_irfft2_bands(spec.data_ptr(), 2 * n, 0, n, B, H, W, n, masks.data_ptr(), Q, workq.data_ptr(), bvs, st), "irfft2_bands")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # and against torch
    sp = torch.complex(spec[..., n:], spec[..., :n])                    # [imag | real] packing
    tref = torch.fft.irfft2(sp * masks[0][None, :, :, None], s=(H, W), dim=(1, 2))
    assert float((out[0] - tref).abs().max()) <= 2e-5 * max(1.0, float(tref.abs().max()))
