// Training path: one DivEnh band of MultiFreq_Refinment with its running sums (reference CVSR_freq.py:2104-2133 applied at :2201-2254,
// bands i >= 1 of the reversed band list):
//     t  = f - Sf + 0.2 So;   e1 = (0.2 a t + b) f;   e2 = (0.2 a So + b) f;   o = e1 * CA(e1) + e2 * CA(e2);   Sf' = Sf + f;   So' = So + o
//     CA(e)[b][c] = sigmoid( W2 relu( W1 mean_HW(e) ) )       (CALayer :1812-1828, the SAME weights for both applications)
// forward in three launches, backward in four (three passes + one launch for the four final row sums), on dense (B, HW, C) f32 tensors (C = 32 or 64, C / 16 hidden
// units).  Under autograd a band was ~30 forward and ~55 backward torch kernels (broadcast multiplies, means, tiny matmuls).
// Backward (gSf', gSo' given; g_o = gSo'):
//   gg_k[b][c] = sum_p g_o e_k;  through the gate: gu = gg g (1 - g), gz = (W2^T gu) [z > 0], gm = W1^T gz, dW2 += gu (x) z, dW1 += gz (x) m;
//   ge_k = g_o g_k + gm_k / HW;   gt = 0.2 a f ge1;
//   gf = gSf' + (0.2 a t + b) ge1 + (0.2 a So + b) ge2 + gt;   gSf = gSf' - gt;   gSo = gSo' + 0.2 gt + 0.2 a f ge2;
//   ga[c] = sum 0.2 f (t ge1 + So ge2);   gb[c] = sum f (ge1 + ge2).
// Every reduction is two-stage with a fixed order (bit-reproducible, no float atomics).
#include "common.h"

namespace fcvsr {

constexpr int kDvPB = 256;              // pixels per block
constexpr int kDvMaxC = 64;

struct DvStats {                         // per image, floats: m1[C] m2[C] g1[C] g2[C] z1[CR] z2[CR]
  static __host__ __device__ int elems(int C) { return 4 * C + 2 * (C / 16); }
};

// thread = (pixel slot of 16, 4 channels); C / 4 lanes per pixel.  MODE 0: sums of e1, e2 (forward means); MODE 1: sums of g e1, g e2
template <int MODE>
__global__ __launch_bounds__(256) void dvb_reduce_kernel(const float* __restrict__ f, const float* __restrict__ sf, const float* __restrict__ so,
                                                         const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                                         int HW, int C, float* __restrict__ part) {
  const int CQ = C / 4, nslot = 256 / CQ;
  const int q = threadIdx.x % CQ, slot = threadIdx.x / CQ;
  const int bi = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const float4 a4 = reinterpret_cast<const float4*>(a)[q], b4 = reinterpret_cast<const float4*>(b)[q];
  const long long base = (long long)bi * HW * C;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int p = blk * kDvPB + slot; p < (blk + 1) * kDvPB && p < HW; p += nslot) {
    const long long o = base + (long long)p * C;
    const float4 fv = reinterpret_cast<const float4*>(f + o)[q], sfv = reinterpret_cast<const float4*>(sf + o)[q],
                 sov = reinterpret_cast<const float4*>(so + o)[q];
    float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (MODE == 1) gv = reinterpret_cast<const float4*>(g + o)[q];
#define FCVSR_DV_E(X)                                                                          \
    {                                                                                          \
      const float t = fv.X - sfv.X + 0.2f * sov.X;                                             \
      s1.X += gv.X * ((0.2f * a4.X * t + b4.X) * fv.X);                                        \
      s2.X += gv.X * ((0.2f * a4.X * sov.X + b4.X) * fv.X);                                    \
    }
    FCVSR_DV_E(x) FCVSR_DV_E(y) FCVSR_DV_E(z) FCVSR_DV_E(w)
#undef FCVSR_DV_E
  }
  __shared__ float sm[2][16 * kDvMaxC];                       // [which][slot][C], nslot * C = 1024
  reinterpret_cast<float4*>(&sm[0][slot * C])[q] = s1;
  reinterpret_cast<float4*>(&sm[1][slot * C])[q] = s2;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int w = i / C, c = i % C;
    float t = 0.f;
    for (int s_ = 0; s_ < nslot; ++s_) t += sm[w][s_ * C + c];
    part[(((long long)bi * nblk + blk) * 2 + w) * C + c] = t;
  }
}

// per image: means of e1 / e2 and the two gates -> stats
__global__ __launch_bounds__(64) void dvb_finish_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ w1, const float* __restrict__ w2,
                                                        int HW, int C, float* __restrict__ stats) {
  const int bi = blockIdx.x, c = threadIdx.x, CR = C / 16;
  __shared__ float m[2][kDvMaxC], z[2][4];
  float* st = stats + (long long)bi * DvStats::elems(C);
  if (c < C) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      float s = 0.f;
      for (int i = 0; i < nblk; ++i) s += part[(((long long)bi * nblk + i) * 2 + w) * C + c];
      m[w][c] = s / (float)HW;
      st[w * C + c] = m[w][c];
    }
  }
  __syncthreads();
  if (c < 2 * CR) {
    const int w = c / CR, h = c % CR;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s += w1[h * C + k] * m[w][k];
    z[w][h] = s > 0.f ? s : 0.f;
    st[4 * C + w * CR + h] = z[w][h];
  }
  __syncthreads();
  if (c < C) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      float s = 0.f;
      for (int h = 0; h < CR; ++h) s += w2[c * CR + h] * z[w][h];
      st[2 * C + w * C + c] = 1.f / (1.f + expf(-s));
    }
  }
}

// Sf' = Sf + f;  So' = So + e1 g1 + e2 g2
__global__ __launch_bounds__(256) void dvb_apply_kernel(const float4* __restrict__ f, const float4* __restrict__ sf, const float4* __restrict__ so,
                                                        const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ stats,
                                                        int HW, int C, float4* __restrict__ sf_out, float4* __restrict__ so_out) {
  const int bi = blockIdx.y, CQ = C / 4;
  const float* st = stats + (long long)bi * DvStats::elems(C);
  const long long n4 = (long long)HW * CQ;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % CQ);
    const float4 a4 = reinterpret_cast<const float4*>(a)[q], b4 = reinterpret_cast<const float4*>(b)[q];
    const float4 g1 = reinterpret_cast<const float4*>(st + 2 * C)[q], g2 = reinterpret_cast<const float4*>(st + 3 * C)[q];
    const long long o = (long long)bi * n4 + i;
    const float4 fv = f[o], sfv = sf[o], sov = so[o];
    float4 nf, no;
#define FCVSR_DV_A(X)                                                                          \
    {                                                                                          \
      const float t = fv.X - sfv.X + 0.2f * sov.X;                                             \
      const float e1 = (0.2f * a4.X * t + b4.X) * fv.X, e2 = (0.2f * a4.X * sov.X + b4.X) * fv.X; \
      nf.X = sfv.X + fv.X;                                                                     \
      no.X = sov.X + (e1 * g1.X + e2 * g2.X);                                                  \
    }
    FCVSR_DV_A(x) FCVSR_DV_A(y) FCVSR_DV_A(z) FCVSR_DV_A(w)
#undef FCVSR_DV_A
    sf_out[o] = nf;
    so_out[o] = no;
  }
}

// per image: gate backward.  part = sums of g_o e1, g_o e2.  bst[b] = {gm1[C], gm2[C]} (already divided by HW); dw1b[b] (CR x C), dw2b[b] (C x CR)
__global__ __launch_bounds__(64) void dvb_bwdmid_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ w1, const float* __restrict__ w2,
                                                        const float* __restrict__ stats, int HW, int C, float* __restrict__ bst,
                                                        float* __restrict__ dw1b, float* __restrict__ dw2b) {
  const int bi = blockIdx.x, c = threadIdx.x, CR = C / 16;
  const float* st = stats + (long long)bi * DvStats::elems(C);
  __shared__ float gu[2][kDvMaxC], gz[2][4];
  if (c < C) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      float s = 0.f;
      for (int i = 0; i < nblk; ++i) s += part[(((long long)bi * nblk + i) * 2 + w) * C + c];
      const float g = st[2 * C + w * C + c];
      gu[w][c] = s * g * (1.f - g);
    }
  }
  __syncthreads();
  if (c < 2 * CR) {
    const int w = c / CR, h = c % CR;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s += w2[k * CR + h] * gu[w][k];
    gz[w][h] = st[4 * C + w * CR + h] > 0.f ? s : 0.f;
  }
  __syncthreads();
  if (c < C) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      float s = 0.f;
      for (int h = 0; h < CR; ++h) s += w1[h * C + c] * gz[w][h];
      bst[(long long)bi * 2 * C + w * C + c] = s / (float)HW;
    }
    // dW2[c][h] = sum_w gu_w[c] z_w[h];  dW1[h][c] = sum_w gz_w[h] m_w[c]
    for (int h = 0; h < CR; ++h) {
      dw2b[((long long)bi * C + c) * CR + h] = gu[0][c] * st[4 * C + h] + gu[1][c] * st[4 * C + CR + h];
      dw1b[((long long)bi * CR + h) * C + c] = gz[0][h] * st[c] + gz[1][h] * st[C + c];
    }
  }
}

// elementwise gradients + per-block partial sums for ga, gb
__global__ __launch_bounds__(256) void dvb_bwd2_kernel(const float* __restrict__ f, const float* __restrict__ sf, const float* __restrict__ so,
                                                       const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ stats,
                                                       const float* __restrict__ bst, const float* __restrict__ gsf, const float* __restrict__ gso,
                                                       int HW, int C, float* __restrict__ gf, float* __restrict__ gsf_out, float* __restrict__ gso_out,
                                                       float* __restrict__ part) {
  const int CQ = C / 4, nslot = 256 / CQ;
  const int q = threadIdx.x % CQ, slot = threadIdx.x / CQ;
  const int bi = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const float* st = stats + (long long)bi * DvStats::elems(C);
  const float4 a4 = reinterpret_cast<const float4*>(a)[q], b4 = reinterpret_cast<const float4*>(b)[q];
  const float4 g1 = reinterpret_cast<const float4*>(st + 2 * C)[q], g2 = reinterpret_cast<const float4*>(st + 3 * C)[q];
  const float4 m1 = reinterpret_cast<const float4*>(bst + (long long)bi * 2 * C)[q], m2 = reinterpret_cast<const float4*>(bst + (long long)bi * 2 * C + C)[q];
  const long long base = (long long)bi * HW * C;
  float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
  for (int p = blk * kDvPB + slot; p < (blk + 1) * kDvPB && p < HW; p += nslot) {
    const long long o = base + (long long)p * C;
    const float4 fv = reinterpret_cast<const float4*>(f + o)[q], sfv = reinterpret_cast<const float4*>(sf + o)[q],
                 sov = reinterpret_cast<const float4*>(so + o)[q];
    const float4 hsf = reinterpret_cast<const float4*>(gsf + o)[q], hso = reinterpret_cast<const float4*>(gso + o)[q];
    float4 of, osf, oso;
#define FCVSR_DV_B(X)                                                                          \
    {                                                                                          \
      const float t = fv.X - sfv.X + 0.2f * sov.X;                                             \
      const float ge1 = hso.X * g1.X + m1.X, ge2 = hso.X * g2.X + m2.X;                        \
      const float gt = 0.2f * a4.X * fv.X * ge1;                                               \
      of.X = hsf.X + (0.2f * a4.X * t + b4.X) * ge1 + (0.2f * a4.X * sov.X + b4.X) * ge2 + gt; \
      osf.X = hsf.X - gt;                                                                      \
      oso.X = hso.X + 0.2f * gt + 0.2f * a4.X * fv.X * ge2;                                    \
      sa.X += 0.2f * fv.X * (t * ge1 + sov.X * ge2);                                           \
      sb.X += fv.X * (ge1 + ge2);                                                              \
    }
    FCVSR_DV_B(x) FCVSR_DV_B(y) FCVSR_DV_B(z) FCVSR_DV_B(w)
#undef FCVSR_DV_B
    reinterpret_cast<float4*>(gf + o)[q] = of;
    reinterpret_cast<float4*>(gsf_out + o)[q] = osf;
    reinterpret_cast<float4*>(gso_out + o)[q] = oso;
  }
  __shared__ float sm[2][16 * kDvMaxC];
  reinterpret_cast<float4*>(&sm[0][slot * C])[q] = sa;
  reinterpret_cast<float4*>(&sm[1][slot * C])[q] = sb;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int w = i / C, c = i % C;
    float t = 0.f;
    for (int s_ = 0; s_ < nslot; ++s_) t += sm[w][s_ * C + c];
    part[(((long long)bi * nblk + blk) * 2 + w) * C + c] = t;
  }
}

// The four final row sums in one launch, fixed order: block 0 -> ga, 1 -> gb (C outputs over B * nblk rows of the interleaved
// [row][2][C] partials: 4 waves x every 4th row, eight loads in flight, wave sums added in order), 2 -> dw1, 3 -> dw2 (C * C/16 outputs,
// B rows).
__global__ __launch_bounds__(256) void dvb_rowsum4_kernel(const float* __restrict__ part, int rows, int C, float* __restrict__ ga,
                                                          float* __restrict__ gb, const float* __restrict__ dw1b, const float* __restrict__ dw2b,
                                                          int B, float* __restrict__ dw1, float* __restrict__ dw2, int accumulate) {
  const int which = blockIdx.x;
  if (which < 2) {
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    __shared__ float sm[4][64];
    float s = 0.f;
    if (o < C) {
      const float* p = part + which * C + o;
      const long long st = 2ll * C;
      int j = g;
      for (; j + 28 < rows; j += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(j + 4 * u) * st];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; j < rows; j += 4) s += p[(long long)j * st];
    }
    sm[g][o] = s;
    __syncthreads();
    if (g == 0 && o < C) {
      float* out = which ? gb : ga;
      const float t = ((sm[0][o] + sm[1][o]) + sm[2][o]) + sm[3][o];
      out[o] = accumulate ? out[o] + t : t;
    }
    return;
  }
  const int n = C * (C / 16);
  const float* in = which == 2 ? dw1b : dw2b;
  float* out = which == 2 ? dw1 : dw2;
  for (int i = threadIdx.x; i < n; i += 256) {
    float s = 0.f;
    for (int j = 0; j < B; ++j) s += in[(long long)j * n + i];
    out[i] = accumulate ? out[i] + s : s;
  }
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_divenh_band_nblk(int HW) { return (HW + kDvPB - 1) / kDvPB; }
extern "C" int fcvsr_divenh_band_stat_elems(int C) { return DvStats::elems(C); }

static bool dv_ok(const void* p) { return p && ((uintptr_t)p % 16) == 0; }

/* forward: f, sf, so, sf_out, so_out dense (B, HW, C) f32; a, b: C floats; w1: (C/16, C), w2: (C, C/16) row-major;
 * stats: B * fcvsr_divenh_band_stat_elems(C) floats (saved for the backward); scratch >= B * nblk * 2 * C floats */
extern "C" int fcvsr_divenh_band_forward(const float* f, const float* sf, const float* so, const float* a, const float* b, const float* w1,
                                         const float* w2, int B, int HW, int C, float* sf_out, float* so_out, float* stats, float* scratch,
                                         long long scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(dv_ok(f) && dv_ok(sf) && dv_ok(so) && dv_ok(a) && dv_ok(b) && w1 && w2 && dv_ok(sf_out) && dv_ok(so_out) && stats && scratch,
                  "null or unaligned pointer");
  FCVSR_CHECK_ARG((C == 32 || C == 64) && B >= 1 && HW >= 1, "C in {32, 64}");
  const int nblk = (HW + kDvPB - 1) / kDvPB;
  FCVSR_CHECK_ARG(scratch_elems >= (long long)B * nblk * 2 * C, "scratch too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dvb_reduce_kernel<0>, dim3(nblk, B), dim3(256), 0, st, f, sf, so, a, b, (const float*)nullptr, HW, C, scratch);
  hipLaunchKernelGGL(dvb_finish_kernel, dim3(B), dim3(64), 0, st, scratch, nblk, w1, w2, HW, C, stats);
  const long long n4 = (long long)HW * (C / 4);
  const int gx = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
  hipLaunchKernelGGL(dvb_apply_kernel, dim3(gx, B), dim3(256), 0, st, (const float4*)f, (const float4*)sf, (const float4*)so, a, b, stats, HW, C,
                     (float4*)sf_out, (float4*)so_out);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

/* backward: gsf, gso = dL/dSf', dL/dSo' -> gf, gsf_out, gso_out (dense) and ga[C], gb[C], dw1[(C/16) x C], dw2[C x (C/16)] (written, or
 * added to when accumulate = 1); scratch >= B * nblk * 2 * C + B * 2 * C + 2 * B * C * (C/16) floats */
extern "C" int fcvsr_divenh_band_backward(const float* f, const float* sf, const float* so, const float* a, const float* b, const float* w1,
                                          const float* w2, const float* stats, const float* gsf, const float* gso, int B, int HW, int C,
                                          float* gf, float* gsf_out, float* gso_out, float* ga, float* gb, float* dw1, float* dw2,
                                          float* scratch, long long scratch_elems, int accumulate, void* stream) {
  FCVSR_CHECK_ARG(dv_ok(f) && dv_ok(sf) && dv_ok(so) && dv_ok(a) && dv_ok(b) && w1 && w2 && stats && dv_ok(gsf) && dv_ok(gso) && dv_ok(gf) &&
                      dv_ok(gsf_out) && dv_ok(gso_out) && ga && gb && dw1 && dw2 && dv_ok(scratch), "null or unaligned pointer");
  FCVSR_CHECK_ARG((C == 32 || C == 64) && B >= 1 && HW >= 1, "C in {32, 64}");
  const int nblk = (HW + kDvPB - 1) / kDvPB, CR = C / 16;
  const long long need = (long long)B * nblk * 2 * C + (long long)B * 2 * C + 2ll * B * C * CR;
  FCVSR_CHECK_ARG(scratch_elems >= need, "scratch too small");
  hipStream_t st = (hipStream_t)stream;
  float* part = scratch;
  float* bst = part + (long long)B * nblk * 2 * C;
  float* dw1b = bst + (long long)B * 2 * C;
  float* dw2b = dw1b + (long long)B * C * CR;
  hipLaunchKernelGGL(dvb_reduce_kernel<1>, dim3(nblk, B), dim3(256), 0, st, f, sf, so, a, b, gso, HW, C, part);
  hipLaunchKernelGGL(dvb_bwdmid_kernel, dim3(B), dim3(64), 0, st, part, nblk, w1, w2, stats, HW, C, bst, dw1b, dw2b);
  hipLaunchKernelGGL(dvb_bwd2_kernel, dim3(nblk, B), dim3(256), 0, st, f, sf, so, a, b, stats, bst, gsf, gso, HW, C, gf, gsf_out, gso_out, part);
  hipLaunchKernelGGL(dvb_rowsum4_kernel, dim3(4), dim3(256), 0, st, part, B * nblk, C, ga, gb, dw1b, dw2b, B, dw1, dw2, accumulate);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
