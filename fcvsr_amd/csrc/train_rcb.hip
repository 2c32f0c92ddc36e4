// Training path: RCB tail with the ContextBlock (reference CVSR_freq.py:657-701 inside RCB.forward :705-725),
//     R = LeakyReLU_0.2( r + add(r) ) + z,     add = W2 . LeakyReLU_0.2( W1 . ctx ),   ctx[c] = sum_p m_p r[p][c],   m = softmax_p( wmask . r[p] )
// forward in three launches and backward in four, on dense (B, HW, 64) f32 tensors (NHWC).  Under autograd this block was ~15
// forward and ~25 backward torch kernels per pyramid level and BlockRCB (36 blocks x 3 levels per step).
// Backward algebra (g = dL/dR):  gu = g * lrelu'(r + add);  gadd = sum_p gu[p];  ga = W2^T gadd;  gt = ga * lrelu'(t);  gctx = W1^T gt;
//   dW2 = gadd (x) a,  dW1 = gt (x) ctx;   softmax: gm_p = gctx . r[p],  sum_p m_p gm_p = gctx . ctx  (no extra pass),
//   gl_p = m_p (gm_p - gctx . ctx);   gr[p] = gu[p] + m_p gctx + wmask gl_p;   dwmask = sum_p gl_p r[p];   gz = g.
// Every reduction is two-stage with a fixed order: bit-reproducible, no float atomics.
#include "common.h"

namespace fcvsr {

constexpr int kTC = 64;                 // channels (n_features)
constexpr int kTPB = 256;               // pixels per block
constexpr int kTStat = 2 + 4 * kTC;     // per-image statistics: M, S, ctx[64], t[64], a[64] (= lrelu(t)), add[64]

__device__ __forceinline__ float grp16_sum(float v) {          // sum over the 16 lanes that share a pixel
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}

// stage 1 of the softmax pool: per block (m, s, ctx[64]) with ctx relative to the block maximum
__global__ __launch_bounds__(256) void rcbt_partial_kernel(const float* __restrict__ r, const float* __restrict__ wmask, int HW, float* __restrict__ part) {
  const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int q = threadIdx.x & 15, slot = threadIdx.x >> 4;      // 16 lanes x 4 channels per pixel, 16 pixels per pass
  const float4 w4 = reinterpret_cast<const float4*>(wmask)[q];
  const float* rb = r + (long long)b * HW * kTC;
  float m = -3.0e38f, s = 0.f;
  float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < kTPB / 16; ++it) {
    const int p = blk * kTPB + it * 16 + slot;
    const bool ok = p < HW;
    const float4 v = reinterpret_cast<const float4*>(rb + (long long)(ok ? p : 0) * kTC)[q];
    const float l = grp16_sum(v.x * w4.x + v.y * w4.y + v.z * w4.z + v.w * w4.w);
    if (ok) {                                                   // online softmax (uniform over the 16 lanes of the pixel)
      const float mn = fmaxf(m, l);
      const float sc = expf(m - mn), e = expf(l - mn);
      s = s * sc + e;
      c4.x = c4.x * sc + e * v.x; c4.y = c4.y * sc + e * v.y; c4.z = c4.z * sc + e * v.z; c4.w = c4.w * sc + e * v.w;
      m = mn;
    }
  }
  __shared__ float sm_m[16], sm_s[16], sm_c[16][kTC];
  if (q == 0) { sm_m[slot] = m; sm_s[slot] = s; }
  reinterpret_cast<float4*>(&sm_c[slot][0])[q] = c4;
  __syncthreads();
  if (threadIdx.x < kTC) {
    float M = sm_m[0];
    for (int i = 1; i < 16; ++i) M = fmaxf(M, sm_m[i]);
    float S = 0.f, cc = 0.f;
    for (int i = 0; i < 16; ++i) {
      const float sc = expf(sm_m[i] - M);
      S += sm_s[i] * sc;
      cc += sm_c[i][threadIdx.x] * sc;
    }
    float* o = part + ((long long)b * nblk + blk) * (kTC + 2);
    if (threadIdx.x == 0) { o[0] = M; o[1] = S; }
    o[2 + threadIdx.x] = cc;
  }
}

// stage 2: per image, combine the partials and run the bottleneck: stats[b] = {M, S, ctx, t = W1 ctx, a = lrelu(t), add = W2 a}
__global__ __launch_bounds__(64) void rcbt_finish_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ w1, const float* __restrict__ w2,
                                                         float slope, float* __restrict__ stats) {
  const int b = blockIdx.x, c = threadIdx.x;
  const float* pb = part + (long long)b * nblk * (kTC + 2);
  float M = -3.0e38f;
  for (int i = 0; i < nblk; ++i) M = fmaxf(M, pb[i * (kTC + 2)]);
  float S = 0.f, cc = 0.f;
  for (int i = 0; i < nblk; ++i) {
    const float sc = expf(pb[i * (kTC + 2)] - M);
    S += pb[i * (kTC + 2) + 1] * sc;
    cc += pb[i * (kTC + 2) + 2 + c] * sc;
  }
  __shared__ float ctx[kTC], av[kTC];
  const float cx = cc / S;
  ctx[c] = cx;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < kTC; ++k) t += w1[c * kTC + k] * ctx[k];
  const float a = t > 0.f ? t : t * slope;
  av[c] = a;
  __syncthreads();
  float ad = 0.f;
  for (int k = 0; k < kTC; ++k) ad += w2[c * kTC + k] * av[k];
  float* st = stats + (long long)b * kTStat;
  if (c == 0) { st[0] = M; st[1] = S; }
  st[2 + c] = cx; st[2 + kTC + c] = t; st[2 + 2 * kTC + c] = a; st[2 + 3 * kTC + c] = ad;
}

// R = lrelu(r + add) + z
__global__ __launch_bounds__(256) void rcbt_apply_kernel(const float4* __restrict__ r, const float4* __restrict__ z, const float* __restrict__ stats,
                                                         float slope, int HW, float4* __restrict__ out) {
  const int b = blockIdx.y;
  const float4* add4 = reinterpret_cast<const float4*>(stats + (long long)b * kTStat + 2 + 3 * kTC);
  const long long n4 = (long long)HW * (kTC / 4);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 a = add4[i & 15], v = r[(long long)b * n4 + i], zz = z[(long long)b * n4 + i];
    float4 u = make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
    u.x = (u.x > 0.f ? u.x : u.x * slope) + zz.x; u.y = (u.y > 0.f ? u.y : u.y * slope) + zz.y;
    u.z = (u.z > 0.f ? u.z : u.z * slope) + zz.z; u.w = (u.w > 0.f ? u.w : u.w * slope) + zz.w;
    out[(long long)b * n4 + i] = u;
  }
}

// backward pass 1: per-block partial of gadd[c] = sum_p g[p][c] * lrelu'(r[p][c] + add[c])
__global__ __launch_bounds__(256) void rcbt_bwd1_kernel(const float* __restrict__ r, const float* __restrict__ g, const float* __restrict__ stats, float slope,
                                                        int HW, float* __restrict__ part) {
  const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int q = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const float4 a4 = reinterpret_cast<const float4*>(stats + (long long)b * kTStat + 2 + 3 * kTC)[q];
  const long long base = (long long)b * HW * kTC;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < kTPB / 16; ++it) {
    const int p = blk * kTPB + it * 16 + slot;
    if (p < HW) {
      const float4 v = reinterpret_cast<const float4*>(r + base + (long long)p * kTC)[q];
      const float4 gg = reinterpret_cast<const float4*>(g + base + (long long)p * kTC)[q];
      acc.x += (v.x + a4.x > 0.f) ? gg.x : gg.x * slope; acc.y += (v.y + a4.y > 0.f) ? gg.y : gg.y * slope;
      acc.z += (v.z + a4.z > 0.f) ? gg.z : gg.z * slope; acc.w += (v.w + a4.w > 0.f) ? gg.w : gg.w * slope;
    }
  }
  __shared__ float sm[16][kTC];
  reinterpret_cast<float4*>(&sm[slot][0])[q] = acc;
  __syncthreads();
  if (threadIdx.x < kTC) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += sm[i][threadIdx.x];
    part[((long long)b * nblk + blk) * kTC + threadIdx.x] = t;
  }
}

// backward middle: per image gadd -> gctx and gctx . ctx; per-image weight gradient terms into dW[b] (summed over b afterwards)
// bst[b] = {gctx[64], gdot}
__global__ __launch_bounds__(64) void rcbt_bwdmid_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ w1, const float* __restrict__ w2,
                                                         const float* __restrict__ stats, float slope, float* __restrict__ bst,
                                                         float* __restrict__ dw1b, float* __restrict__ dw2b) {
  const int b = blockIdx.x, c = threadIdx.x;
  const float* st = stats + (long long)b * kTStat;
  float gadd = 0.f;
  for (int i = 0; i < nblk; ++i) gadd += part[((long long)b * nblk + i) * kTC + c];
  __shared__ float s_gadd[kTC], s_gt[kTC], s_gc[kTC];
  s_gadd[c] = gadd;
  __syncthreads();
  float ga = 0.f;                                               // ga[c] = sum_k W2[k][c] gadd[k]
  for (int k = 0; k < kTC; ++k) ga += w2[k * kTC + c] * s_gadd[k];
  const float t = st[2 + kTC + c];
  const float gt = t > 0.f ? ga : ga * slope;
  s_gt[c] = gt;
  __syncthreads();
  float gc = 0.f;                                               // gctx[c] = sum_k W1[k][c] gt[k]
  for (int k = 0; k < kTC; ++k) gc += w1[k * kTC + c] * s_gt[k];
  s_gc[c] = gc * st[2 + c];
  bst[(long long)b * (kTC + 1) + c] = gc;
  __syncthreads();
  if (c == 0) {
    float d = 0.f;
    for (int k = 0; k < kTC; ++k) d += s_gc[k];
    bst[(long long)b * (kTC + 1) + kTC] = d;
  }
  // dW2[c][k] = gadd[c] * a[k];  dW1[c][k] = gt[c] * ctx[k]
  float* o1 = dw1b + (long long)b * kTC * kTC, *o2 = dw2b + (long long)b * kTC * kTC;
  // (loop vectorisation off: it forms packed-FP32 multiplies, which no code object of this library may contain - build.py)
#pragma clang loop vectorize(disable) interleave(disable)
  for (int k = 0; k < kTC; ++k) {
    o2[c * kTC + k] = gadd * st[2 + 2 * kTC + k];
    o1[c * kTC + k] = gt * st[2 + k];
  }
}

// backward pass 2: gr and the per-block partial of dwmask[c] = sum_p gl_p r[p][c]
__global__ __launch_bounds__(256) void rcbt_bwd2_kernel(const float* __restrict__ r, const float* __restrict__ g, const float* __restrict__ wmask,
                                                        const float* __restrict__ stats, const float* __restrict__ bst, float slope, int HW,
                                                        float* __restrict__ gr, float* __restrict__ part) {
  const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int q = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const float* st = stats + (long long)b * kTStat;
  const float M = st[0], invS = 1.f / st[1];
  const float4 a4 = reinterpret_cast<const float4*>(st + 2 + 3 * kTC)[q];
  const float4 w4 = reinterpret_cast<const float4*>(wmask)[q];
  const float4 gc4 = reinterpret_cast<const float4*>(bst + (long long)b * (kTC + 1))[q];
  const float gdot = bst[(long long)b * (kTC + 1) + kTC];
  const long long base = (long long)b * HW * kTC;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < kTPB / 16; ++it) {
    const int p = blk * kTPB + it * 16 + slot;
    const bool ok = p < HW;
    const long long off = base + (long long)(ok ? p : 0) * kTC;
    const float4 v = reinterpret_cast<const float4*>(r + off)[q];
    const float4 gg = reinterpret_cast<const float4*>(g + off)[q];
    const float l = grp16_sum(v.x * w4.x + v.y * w4.y + v.z * w4.z + v.w * w4.w);
    const float gm = grp16_sum(v.x * gc4.x + v.y * gc4.y + v.z * gc4.z + v.w * gc4.w);
    const float mp = expf(l - M) * invS;
    const float gl = mp * (gm - gdot);
    if (ok) {
      float4 o;
      o.x = ((v.x + a4.x > 0.f) ? gg.x : gg.x * slope) + mp * gc4.x + w4.x * gl;
      o.y = ((v.y + a4.y > 0.f) ? gg.y : gg.y * slope) + mp * gc4.y + w4.y * gl;
      o.z = ((v.z + a4.z > 0.f) ? gg.z : gg.z * slope) + mp * gc4.z + w4.z * gl;
      o.w = ((v.w + a4.w > 0.f) ? gg.w : gg.w * slope) + mp * gc4.w + w4.w * gl;
      reinterpret_cast<float4*>(gr + off)[q] = o;
      acc.x += gl * v.x; acc.y += gl * v.y; acc.z += gl * v.z; acc.w += gl * v.w;
    }
  }
  __shared__ float sm[16][kTC];
  reinterpret_cast<float4*>(&sm[slot][0])[q] = acc;
  __syncthreads();
  if (threadIdx.x < kTC) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += sm[i][threadIdx.x];
    part[((long long)b * nblk + blk) * kTC + threadIdx.x] = t;
  }
}

// The three final row sums of the backward pass in ONE launch (they were three: 108 launches per training step, the first a single
// workgroup walking up to 256 rows one load at a time, 60 us): block 0 -> dwmask[64] over B * nblk rows (4 waves x every 4th row,
// eight loads in flight, wave sums added in order), blocks 1 .. 16 -> dw1, 17 .. 32 -> dw2 (4096 outputs, B rows each).  Fixed order.
__global__ __launch_bounds__(256) void rcbt_rowsum3_kernel(const float* __restrict__ part, int rows_m, float* __restrict__ dwmask,
                                                           const float* __restrict__ dw1b, const float* __restrict__ dw2b, int B,
                                                           float* __restrict__ dw1, float* __restrict__ dw2, int accumulate) {
  if (blockIdx.x == 0) {
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    __shared__ float sm[4][64];
    float s = 0.f;
    int j = g;
    for (; j + 28 < rows_m; j += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long long)(j + 4 * u) * kTC + o];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; j < rows_m; j += 4) s += part[(long long)j * kTC + o];
    sm[g][o] = s;
    __syncthreads();
    if (g == 0) {
      const float t = ((sm[0][o] + sm[1][o]) + sm[2][o]) + sm[3][o];
      dwmask[o] = accumulate ? dwmask[o] + t : t;
    }
    return;
  }
  const int blk = blockIdx.x - 1, which = blk / (kTC * kTC / 256);
  const int i = (blk % (kTC * kTC / 256)) * 256 + threadIdx.x;
  const float* in = which ? dw2b : dw1b;
  float* out = which ? dw2 : dw1;
  float s = 0.f;
  for (int j = 0; j < B; ++j) s += in[(long long)j * (kTC * kTC) + i];
  out[i] = accumulate ? out[i] + s : s;
}

// Adjoint of the x2 bilinear up-sampling of fcvsr_xscale (align_corners = False, source index clamped at 0 and at the last row /
// column): low-resolution pixel (Y, X) collects from the high-resolution rows 2Y-1 .. 2Y+2 with weights 0.25 / 0.75 / 0.75 / 0.25,
// the clamped share of the first and last row staying on the edge pixel; same along x.  g: (B, 2H, 2W, C), out: (B, H, W, C), f32.
__global__ __launch_bounds__(256) void up2_adjoint_kernel(const float4* __restrict__ g, float4* __restrict__ out, int B, int H, int W, int Cq) {
  const long long total = (long long)B * H * W * Cq;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int cq = (int)(t % Cq);
  const long long pg = t / Cq;
  const int X = (int)(pg % W), Y = (int)((pg / W) % H), b = (int)(pg / ((long long)W * H));
  const int H2 = 2 * H, W2 = 2 * W;
  float wy[4], wx[4];
  int iy[4], ix[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    iy[k] = 2 * Y - 1 + k; ix[k] = 2 * X - 1 + k;
    wy[k] = (k == 0 || k == 3) ? 0.25f : 0.75f;
    wx[k] = wy[k];
  }
  // edges: the up-sampling clamps its source index, so row 0 of the output reads low-res row 0 with weight 1 (and so does row 2H-1
  // with low-res row H-1); rows outside [0, 2H) do not exist
  if (Y == 0) { wy[0] = 0.f; wy[1] = 1.f; }
  if (Y == H - 1) { wy[3] = 0.f; wy[2] = 1.f; }
  if (X == 0) { wx[0] = 0.f; wx[1] = 1.f; }
  if (X == W - 1) { wx[3] = 0.f; wx[2] = 1.f; }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (wy[a] == 0.f) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (wx[c] == 0.f) continue;
      const float w = wy[a] * wx[c];
      const float4 v = g[(((long long)b * H2 + iy[a]) * W2 + ix[c]) * Cq + cq];
      acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
  }
  out[t] = acc;
}

// Adjoint of the 2x2 mean: out (B, 2H, 2W, C)[i][j] = 0.25 * g (B, H, W, C)[i / 2][j / 2]
__global__ __launch_bounds__(256) void pool2_adjoint_kernel(const float4* __restrict__ g, float4* __restrict__ out, int B, int H, int W, int Cq) {
  const long long total = (long long)B * (2 * H) * (2 * W) * Cq;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int cq = (int)(t % Cq);
  const long long pg = t / Cq;
  const int j = (int)(pg % (2 * W)), i = (int)((pg / (2 * W)) % (2 * H)), b = (int)(pg / ((long long)4 * W * H));
  const float4 v = g[(((long long)b * H + (i >> 1)) * W + (j >> 1)) * Cq + cq];
  out[t] = make_float4(0.25f * v.x, 0.25f * v.y, 0.25f * v.z, 0.25f * v.w);
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_rcbt_nblk(int HW) { return (HW + kTPB - 1) / kTPB; }
extern "C" int fcvsr_rcbt_stat_elems(void) { return kTStat; }

/* forward: r, z, out dense (B, HW, 64) f32; stats (B, fcvsr_rcbt_stat_elems()) is saved for the backward;
 * scratch >= B * nblk * 66 floats */
extern "C" int fcvsr_rcbt_forward(const float* r, const float* z, const float* wmask, const float* w1, const float* w2, float slope,
                                  int B, int HW, int C, float* out, float* stats, float* scratch, long long scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(r && z && wmask && w1 && w2 && out && stats && scratch, "null pointer");
  FCVSR_CHECK_ARG(C == kTC, "64 channels");
  FCVSR_CHECK_ARG(B >= 1 && HW >= 1, "empty tensor");
  const int nblk = (HW + kTPB - 1) / kTPB;
  FCVSR_CHECK_ARG(scratch_elems >= (long long)B * nblk * (kTC + 2), "scratch too small");
  FCVSR_CHECK_ARG(((uintptr_t)r % 16) == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)wmask % 16) == 0, "16-byte aligned tensors");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rcbt_partial_kernel, dim3(nblk, B), dim3(256), 0, st, r, wmask, HW, scratch);
  hipLaunchKernelGGL(rcbt_finish_kernel, dim3(B), dim3(64), 0, st, scratch, nblk, w1, w2, slope, stats);
  const long long n4 = (long long)HW * (kTC / 4);
  const int gx = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
  hipLaunchKernelGGL(rcbt_apply_kernel, dim3(gx, B), dim3(256), 0, st, (const float4*)r, (const float4*)z, stats, slope, HW, (float4*)out);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

/* backward: g = dL/dout; writes gr (dL/dr; dL/dz = g) and writes (accumulate = 0) or adds to (1) dwmask[64], dw1[64*64], dw2[64*64];
 * scratch >= B * nblk * 64 + B * 65 + 2 * B * 4096 floats */
extern "C" int fcvsr_rcbt_backward(const float* r, const float* g, const float* wmask, const float* w1, const float* w2, const float* stats,
                                   float slope, int B, int HW, int C, float* gr, float* dwmask, float* dw1, float* dw2, float* scratch,
                                   long long scratch_elems, int accumulate, void* stream) {
  FCVSR_CHECK_ARG(r && g && wmask && w1 && w2 && stats && gr && dwmask && dw1 && dw2 && scratch, "null pointer");
  FCVSR_CHECK_ARG(C == kTC, "64 channels");
  const int nblk = (HW + kTPB - 1) / kTPB;
  const long long need = (long long)B * nblk * kTC + (long long)B * (kTC + 1) + 2ll * B * kTC * kTC;
  FCVSR_CHECK_ARG(scratch_elems >= need, "scratch too small");
  FCVSR_CHECK_ARG(((uintptr_t)r % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)gr % 16) == 0 && ((uintptr_t)wmask % 16) == 0, "16-byte aligned tensors");
  hipStream_t st = (hipStream_t)stream;
  float* part = scratch;
  float* bst = part + (long long)B * nblk * kTC;                // (kTC + 1) per image: keep 16-byte alignment of the rows below
  float* dw1b = bst + (((long long)B * (kTC + 1) + 3) / 4) * 4;
  float* dw2b = dw1b + (long long)B * kTC * kTC;
  FCVSR_CHECK_ARG(scratch_elems >= (dw2b - scratch) + (long long)B * kTC * kTC, "scratch too small");
  hipLaunchKernelGGL(rcbt_bwd1_kernel, dim3(nblk, B), dim3(256), 0, st, r, g, stats, slope, HW, part);
  hipLaunchKernelGGL(rcbt_bwdmid_kernel, dim3(B), dim3(64), 0, st, part, nblk, w1, w2, stats, slope, bst, dw1b, dw2b);
  hipLaunchKernelGGL(rcbt_bwd2_kernel, dim3(nblk, B), dim3(256), 0, st, r, g, wmask, stats, bst, slope, HW, gr, part);
  hipLaunchKernelGGL(rcbt_rowsum3_kernel, dim3(1 + 2 * (kTC * kTC / 256)), dim3(256), 0, st, part, B * nblk, dwmask, dw1b, dw2b, B, dw1, dw2,
                     accumulate);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

/* g (B, 2H, 2W, C) -> out (B, H, W, C): adjoint of the x2 bilinear up-sampling inside fcvsr_xscale */
extern "C" int fcvsr_up2_adjoint(const float* g, float* out, int B, int H, int W, int C, void* stream) {
  FCVSR_CHECK_ARG(g && out && B > 0 && H > 0 && W > 0 && C % 4 == 0, "bad arguments");
  FCVSR_CHECK_ARG(((uintptr_t)g % 16) == 0 && ((uintptr_t)out % 16) == 0, "16-byte aligned tensors");
  const long long total = (long long)B * H * W * (C / 4);
  hipLaunchKernelGGL(up2_adjoint_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)g, (float4*)out, B, H, W, C / 4);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

/* g (B, H, W, C) -> out (B, 2H, 2W, C): adjoint of the 2x2 mean inside fcvsr_xscale */
extern "C" int fcvsr_pool2_adjoint(const float* g, float* out, int B, int H, int W, int C, void* stream) {
  FCVSR_CHECK_ARG(g && out && B > 0 && H > 0 && W > 0 && C % 4 == 0, "bad arguments");
  FCVSR_CHECK_ARG(((uintptr_t)g % 16) == 0 && ((uintptr_t)out % 16) == 0, "16-byte aligned tensors");
  const long long total = (long long)B * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(pool2_adjoint_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)g, (float4*)out, B, H, W, C / 4);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
