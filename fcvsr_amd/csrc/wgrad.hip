// Weight gradient of a 2-D convolution (the backward half of every nn.Conv2d on the path: reference
// CVSR_train/train_LD_freqCVSR_S_22.py:250 `loss.backward()` through CVSR_freq.py's convolutions).
//
//   dW[co][ci][ky][kx] = sum_{b,oy,ox} gy[b,oy,ox,co] * x[b, oy*stride - pad + ky, ox*stride - pad + kx, ci]
//
// Exact f32 and bit-reproducible: the pixel list is cut into slabs, a workgroup accumulates one (slab, tap, 16-cout x 64-cin
// block) in registers in a fixed order, and a second launch adds the slabs in order (no float atomics).  This is the
// exact-parity path (the reference trains in f32); the 16-bit matrix-core variant below is the throughput path.
//
// Layout: x and gy are channel-contiguous (NHWC) f32 views; the result is written in the reference's parameter layout
// (cout, cin, kh, kw) so it can be returned as the .grad of the nn.Conv2d weight as is.
#include "common.h"

namespace fcvsr {

struct WgradArgs {
  View x, gy;
  int B, H, W, Ho, Wo, kh, kw, stride, pad, cin, cout;
  long long npix;          // B * Ho * Wo
  int n_slabs;
  long long slab_pix;      // pixels per slab
  float* partial;          // [n_slabs][kh*kw][cin][cout]
  float* dw;               // [cout][cin][kh][kw]
};

constexpr int kWgCo = 16, kWgCi = 64, kWgPx = 32;

__global__ __launch_bounds__(256) void wgrad_partial_kernel(WgradArgs a) {
  __shared__ float gy_s[kWgPx][kWgCo];
  __shared__ __align__(16) float x_s[kWgPx][kWgCi];
  const int tid = threadIdx.x;
  const int slab = blockIdx.x, tap = blockIdx.y;
  const int nci = (a.cin + kWgCi - 1) / kWgCi;
  const int co0 = (blockIdx.z / nci) * kWgCo, ci0 = (blockIdx.z % nci) * kWgCi;
  const int ky = tap / a.kw, kx = tap % a.kw;
  const int tco = tid & 15, tci = tid >> 4;                 // 16 couts x 16 groups of 4 cins
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const long long p0 = (long long)slab * a.slab_pix;
  long long p1 = p0 + a.slab_pix;
  if (p1 > a.npix) p1 = a.npix;
  for (long long pc = p0; pc < p1; pc += kWgPx) {
    __syncthreads();
    // stage gy[pc .. pc+32)[co0 .. co0+16) and the tap-shifted x[..][ci0 .. ci0+64) (zeros outside the image / past the slab)
    for (int i = tid; i < kWgPx * kWgCo; i += 256) {
      const int q = i >> 4, c = i & 15;
      const long long p = pc + q;
      float v = 0.f;
      if (p < p1 && co0 + c < a.cout) {
        const int ox = (int)(p % a.Wo);
        const int oy = (int)((p / a.Wo) % a.Ho);
        const int b = (int)(p / ((long long)a.Wo * a.Ho));
        v = a.gy.p[(long long)b * a.gy.sb + (long long)oy * a.gy.sy + (long long)ox * a.gy.sx + co0 + c];
      }
      gy_s[q][c] = v;
    }
    for (int i = tid; i < kWgPx * kWgCi; i += 256) {
      const int q = i >> 6, c = i & 63;
      const long long p = pc + q;
      float v = 0.f;
      if (p < p1 && ci0 + c < a.cin) {
        const int ox = (int)(p % a.Wo);
        const int oy = (int)((p / a.Wo) % a.Ho);
        const int b = (int)(p / ((long long)a.Wo * a.Ho));
        const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
          v = a.x.p[(long long)b * a.x.sb + (long long)iy * a.x.sy + (long long)ix * a.x.sx + ci0 + c];
      }
      x_s[q][c] = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int q = 0; q < kWgPx; ++q) {
      const float g = gy_s[q][tco];
      const float4 xv = *reinterpret_cast<const float4*>(&x_s[q][tci * 4]);
      acc[0] = fmaf(g, xv.x, acc[0]);
      acc[1] = fmaf(g, xv.y, acc[1]);
      acc[2] = fmaf(g, xv.z, acc[2]);
      acc[3] = fmaf(g, xv.w, acc[3]);
    }
  }
  const int co = co0 + tco;
  if (co < a.cout) {
    float* pp = a.partial + (((long long)slab * (a.kh * a.kw) + tap) * a.cin) * a.cout;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ci = ci0 + tci * 4 + e;
      if (ci < a.cin) pp[(long long)ci * a.cout + co] = acc[e];
    }
  }
}

// dw[co][ci][tap] = sum over slabs, in slab order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs a) {
  const long long n = (long long)a.cout * a.cin * a.kh * a.kw;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int taps = a.kh * a.kw;
  const int tap = (int)(i % taps);
  const int ci = (int)((i / taps) % a.cin);
  const int co = (int)(i / ((long long)taps * a.cin));
  float s = 0.f;
  for (int sl = 0; sl < a.n_slabs; ++sl) s += a.partial[(((long long)sl * taps + tap) * a.cin + ci) * a.cout + co];
  a.dw[i] = s;
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" long long fcvsr_conv2d_wgrad_scratch_elems(int B, int Ho, int Wo, int cin, int cout, int kh, int kw) {
  const long long npix = (long long)B * Ho * Wo;
  long long n_slabs = (npix + 511) / 512;
  if (n_slabs > 96) n_slabs = 96;
  if (n_slabs < 1) n_slabs = 1;
  return n_slabs * kh * kw * (long long)cin * cout;
}

extern "C" int fcvsr_conv2d_wgrad(const fcvsr_view* x, const fcvsr_view* gy, int B, int H, int W, int kh, int kw, int stride, int pad,
                                  float* dw, float* scratch, long long scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(x && gy && dw && scratch, "null argument");
  FCVSR_CHECK_ARG(x->dtype == FCVSR_F32 && gy->dtype == FCVSR_F32 && x->sc == 1 && gy->sc == 1 && x->ptr && gy->ptr,
                  "x and gy must be channel-contiguous f32 views");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && kh >= 1 && kw >= 1 && stride >= 1 && pad >= 0, "bad geometry");
  WgradArgs a;
  a.x = to_view(*x);
  a.gy = to_view(*gy);
  a.B = B; a.H = H; a.W = W; a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
  a.Ho = (H + 2 * pad - kh) / stride + 1;
  a.Wo = (W + 2 * pad - kw) / stride + 1;
  FCVSR_CHECK_ARG(a.Ho > 0 && a.Wo > 0, "empty output");
  a.cin = x->c; a.cout = gy->c;
  a.npix = (long long)B * a.Ho * a.Wo;
  const long long need = fcvsr_conv2d_wgrad_scratch_elems(B, a.Ho, a.Wo, a.cin, a.cout, kh, kw);
  FCVSR_CHECK_ARG(scratch_elems >= need, "scratch too small (fcvsr_conv2d_wgrad_scratch_elems)");
  a.n_slabs = (int)(need / ((long long)kh * kw * a.cin * a.cout));
  a.slab_pix = (a.npix + a.n_slabs - 1) / a.n_slabs;
  a.slab_pix = (a.slab_pix + kWgPx - 1) / kWgPx * kWgPx;
  a.partial = scratch;
  a.dw = dw;
  hipStream_t st = (hipStream_t)stream;
  const int nco = (a.cout + kWgCo - 1) / kWgCo, nci = (a.cin + kWgCi - 1) / kWgCi;
  FCVSR_CHECK_ARG(kh * kw <= 65535 && (long long)nco * nci <= 65535, "grid too large");
  hipLaunchKernelGGL(wgrad_partial_kernel, dim3(a.n_slabs, kh * kw, nco * nci), dim3(256), 0, st, a);
  FCVSR_LAUNCH_CHECK();
  const long long n = (long long)a.cout * a.cin * kh * kw;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
