// One ConvBlk head of MGAAbk (reference CVSR_freq.py:344-357, applied at :1494-1498) on the 4-channel offset spectra:
//     u   = conv2( PReLU( conv1(x) ) )                  k x k, 4 -> 4, no bias, k = 2*index+1
//     out = (CA(u) + u) * sim                           CALayer(4, reduction 1, no bias), split into (real, imag) planes
// in two launches instead of six (conv1, conv2, 2-stage channel sums, gate, tail): the tensors are tiny (16 bytes per
// pixel), so the stand-alone launches were pure latency.
//   1. convblk_conv_kernel<K>: 16 x 16 output pixels per workgroup; the input tile with a 2*(K/2) halo and the PReLU'd
//      intermediate with a K/2 halo live in LDS (zero padding of BOTH convolutions reproduced: the intermediate is zero outside
//      the image); weights are wave-uniform scalar loads; per-workgroup channel sums of u in a fixed tree order.
//   2. convblk_tail_kernel2: every workgroup first reduces the partial sums of its batch item (fixed order) and evaluates the
//      4 -> 4 -> 4 gate, then applies it.  Deterministic: no atomics anywhere.
#include "common.h"

namespace fcvsr {

constexpr int kCbT = 16;                                  // output tile side

template <int K>
__global__ __launch_bounds__(256) void convblk_conv_kernel(const float4* x, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, const float* slope_p, float4* u,
                                                           float4* partial, int H, int W, int tiles_x, int tiles_y) {
  constexpr int P = K / 2;
  constexpr int XS = kCbT + 4 * P, TT = kCbT + 2 * P;
  __shared__ float4 xs[XS * XS];
  __shared__ float4 ts[TT * TT];
  __shared__ float4 red[256];
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int ty0 = (blockIdx.x / tiles_x) * kCbT, tx0 = (blockIdx.x % tiles_x) * kCbT;
  const float4* xn = x + (long long)n * H * W;
  const float slope = slope_p[0];
  for (int idx = tid; idx < XS * XS; idx += 256) {
    const int yy = idx / XS, xx = idx - yy * XS;
    const int gy = ty0 - 2 * P + yy, gx = tx0 - 2 * P + xx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = xn[(long long)gy * W + gx];
    xs[idx] = v;
  }
  __syncthreads();
  // weights: [tap][cin][16] f32 (cout padded to 16 by pack_conv_weight); uniform addresses -> scalar loads
  for (int idx = tid; idx < TT * TT; idx += 256) {
    const int yy = idx / TT, xx = idx - yy * TT;
    const int gy = ty0 - P + yy, gx = tx0 - P + xx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float4 v = xs[(yy + ky) * XS + xx + kx];
          const float* wt = w1 + (ky * K + kx) * 64;
          const float vi[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) {
            a0 = fmaf(vi[ci], wt[ci * 16 + 0], a0); a1 = fmaf(vi[ci], wt[ci * 16 + 1], a1);
            a2 = fmaf(vi[ci], wt[ci * 16 + 2], a2); a3 = fmaf(vi[ci], wt[ci * 16 + 3], a3);
          }
        }
      }
      a0 = a0 >= 0.f ? a0 : a0 * slope; a1 = a1 >= 0.f ? a1 : a1 * slope;
      a2 = a2 >= 0.f ? a2 : a2 * slope; a3 = a3 >= 0.f ? a3 : a3 * slope;
    }
    ts[idx] = make_float4(a0, a1, a2, a3);                // zero outside the image = zero padding of conv2
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  const int gy = ty0 + ly, gx = tx0 + lx;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gy < H && gx < W) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float4 v = ts[(ly + ky) * TT + lx + kx];
        const float* wt = w2 + (ky * K + kx) * 64;
        const float vi[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          a0 = fmaf(vi[ci], wt[ci * 16 + 0], a0); a1 = fmaf(vi[ci], wt[ci * 16 + 1], a1);
          a2 = fmaf(vi[ci], wt[ci * 16 + 2], a2); a3 = fmaf(vi[ci], wt[ci * 16 + 3], a3);
        }
      }
    }
    o = make_float4(a0, a1, a2, a3);
    u[((long long)n * H + gy) * W + gx] = o;
  }
  red[tid] = o;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) {
      const float4 p = red[tid], q = red[tid + st];
      red[tid] = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
    }
    __syncthreads();
  }
  if (tid == 0) partial[(long long)n * gridDim.x + blockIdx.x] = red[0];
}

// out = (u*gate + u) * sim -> (real, imag) planes; gate = sigmoid(W2 relu(W1 mean(u)))  (CALayer :1812-1828, C = CR = 4)
__global__ __launch_bounds__(256) void convblk_tail_kernel2(const float4* u, const float4* partial, int nblk, float inv_hw,
                                                            const float* ca_w1, const float* ca_w2, const float4* sim, int B,
                                                            long long HW, float* spec, long long ps, int re_off, int im_off,
                                                            int g_stride, int g0) {
  __shared__ float4 red[256];
  __shared__ float gate_s[4];
  const int tid = threadIdx.x;
  const int bn = blockIdx.y;                              // dir * B + b
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = tid; k < nblk; k += 256) {
    const float4 p = partial[(long long)bn * nblk + k];
    s = make_float4(s.x + p.x, s.y + p.y, s.z + p.z, s.w + p.w);
  }
  red[tid] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) {
      const float4 p = red[tid], q = red[tid + st];
      red[tid] = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
    }
    __syncthreads();
  }
  if (tid < 4) {
    const float4 t = red[0];
    const float mean[4] = {t.x * inv_hw, t.y * inv_hw, t.z * inv_hw, t.w * inv_hw};
    float hid[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) a = fmaf(ca_w1[h * 4 + c], mean[c], a);
      hid[h] = fmaxf(a, 0.f);
    }
    float a = 0.f;
#pragma unroll
    for (int h = 0; h < 4; ++h) a = fmaf(ca_w2[tid * 4 + h], hid[h], a);
    gate_s[tid] = 1.f / (1.f + expf(-a));
  }
  __syncthreads();
  const long long pix = (long long)blockIdx.x * 256 + tid;
  if (pix >= HW) return;
  const int dir = bn / B, b = bn - dir * B;
  const float4 uu = u[(long long)bn * HW + pix];
  const float4 ss = sim[(long long)b * HW + pix];
  const float o0 = fmaf(uu.x, gate_s[0], uu.x) * ss.x;
  const float o1 = fmaf(uu.y, gate_s[1], uu.y) * ss.y;
  const float o2 = fmaf(uu.z, gate_s[2], uu.z) * ss.z;
  const float o3 = fmaf(uu.w, gate_s[3], uu.w) * ss.w;
  float* px = spec + ((long long)b * HW + pix) * ps;
  const int gi = (g0 + dir * g_stride) * 2;
  if (((ps | re_off | im_off) & 1) == 0 && (reinterpret_cast<uintptr_t>(spec) & 7) == 0) {   // gi is even: 8-byte pairs
    *reinterpret_cast<float2*>(px + re_off + gi) = make_float2(o0, o1);
    *reinterpret_cast<float2*>(px + im_off + gi) = make_float2(o2, o3);
  } else {
    px[re_off + gi] = o0;
    px[re_off + gi + 1] = o1;
    px[im_off + gi] = o2;
    px[im_off + gi + 1] = o3;
  }
}

template <int K>
static void launch_convblk_conv(dim3 grid, hipStream_t st, const float4* x, const float* w1, const float* w2, const float* slope,
                                float4* u, float4* partial, int H, int W, int tx, int ty) {
  hipLaunchKernelGGL(convblk_conv_kernel<K>, grid, dim3(256), 0, st, x, w1, w2, slope, u, partial, H, W, tx, ty);
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_convblk(const float* x, const float* w1, const float* w2, const float* prelu_slope, int ksize,
                             const float* ca_w1, const float* ca_w2, const float* sim, int B, int ndir, int H, int Wf,
                             float* u_scratch, float* partial_scratch, int64_t partial_elems, float* spec, int64_t pix_stride,
                             int re_off, int im_off, int g_stride, int g0, void* stream) {
  FCVSR_CHECK_ARG(x && w1 && w2 && prelu_slope && ca_w1 && ca_w2 && sim && u_scratch && partial_scratch && spec, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && ndir > 0 && H > 0 && Wf > 0, "bad sizes");
  FCVSR_CHECK_ARG(ksize % 2 == 1 && ksize >= 1 && ksize <= 11, "kernel size: odd, 1..11");
  FCVSR_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)u_scratch % 16) == 0 && ((uintptr_t)sim % 16) == 0 &&
                      ((uintptr_t)partial_scratch % 16) == 0, "16-byte alignment");
  const int tx = cdiv(Wf, kCbT), ty = cdiv(H, kCbT);
  const int nblk = tx * ty, N = ndir * B;
  FCVSR_CHECK_ARG(partial_elems >= 4ll * N * nblk, "partial scratch too small");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nblk, N);
  const float4* x4 = (const float4*)x;
  float4* u4 = (float4*)u_scratch;
  float4* p4 = (float4*)partial_scratch;
  switch (ksize) {
    case 1: launch_convblk_conv<1>(grid, st, x4, w1, w2, prelu_slope, u4, p4, H, Wf, tx, ty); break;
    case 3: launch_convblk_conv<3>(grid, st, x4, w1, w2, prelu_slope, u4, p4, H, Wf, tx, ty); break;
    case 5: launch_convblk_conv<5>(grid, st, x4, w1, w2, prelu_slope, u4, p4, H, Wf, tx, ty); break;
    case 7: launch_convblk_conv<7>(grid, st, x4, w1, w2, prelu_slope, u4, p4, H, Wf, tx, ty); break;
    case 9: launch_convblk_conv<9>(grid, st, x4, w1, w2, prelu_slope, u4, p4, H, Wf, tx, ty); break;
    default: launch_convblk_conv<11>(grid, st, x4, w1, w2, prelu_slope, u4, p4, H, Wf, tx, ty); break;
  }
  FCVSR_LAUNCH_CHECK();
  const long long HW = (long long)H * Wf;
  hipLaunchKernelGGL(convblk_tail_kernel2, dim3(cdiv(HW, 256), N), dim3(256), 0, st, (const float4*)u4, (const float4*)p4, nblk,
                     1.0f / (float)HW, ca_w1, ca_w2, (const float4*)sim, B, HW, spec, (long long)pix_stride, re_off, im_off,
                     g_stride, g0);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
