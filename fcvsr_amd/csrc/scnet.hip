// SCNetbk non-conv kernels (reference CVSR_freq.py:657-777): ContextBlock global softmax pooling + MLP, RCB tail,
// BlockRCB cross-scale sum (2x2 mean down, bilinear x2 up).  Dense NHWC (B,H,W,C) f32.
#include "common.h"

namespace fcvsr {

constexpr int kGcPix = 256;  // pixels per stage-1 block (= threads)

// Stage 1: per block of 256 pixels: logits l_p = r_p . wmask, m = max l, e_p = exp(l_p - m),
// part[b][blk][0..C) = sum_p e_p r_p[c], part[..][C] = m, part[..][C+1] = sum_p e_p
__global__ __launch_bounds__(kGcPix) void gc_stage1_kernel(const float* r, const float* wmask, long long HW, int C,
                                                           float* part) {
  __shared__ float e_s[kGcPix];
  __shared__ float red[kGcPix];
  const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const long long p = (long long)blk * kGcPix + threadIdx.x;
  const bool live = p < HW;
  float logit = -INFINITY;
  if (live) {
    const float4* rp = reinterpret_cast<const float4*>(r + ((long long)b * HW + p) * C);
    const float4* wp = reinterpret_cast<const float4*>(wmask);
    float s = 0.f;
    for (int q = 0; q < C / 4; ++q) {
      const float4 v = rp[q], w = wp[q];
      s = fmaf(v.x, w.x, s); s = fmaf(v.y, w.y, s); s = fmaf(v.z, w.z, s); s = fmaf(v.w, w.w, s);
    }
    logit = s;
  }
  red[threadIdx.x] = logit;
  __syncthreads();
  for (int st = kGcPix / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  const float m = red[0];
  __syncthreads();
  const float e = live ? expf(logit - m) : 0.f;
  e_s[threadIdx.x] = e;
  red[threadIdx.x] = e;
  __syncthreads();
  for (int st = kGcPix / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  const float esum = red[0];
  __syncthreads();
  // weighted channel sums: thread (sub, c)
  const int R = kGcPix / C;
  const int sub = threadIdx.x / C, c = threadIdx.x % C;
  float acc = 0.f;
  if (sub < R) {
    const long long p0 = (long long)blk * kGcPix;
    const int np = (int)((HW - p0 < kGcPix) ? (HW - p0) : kGcPix);
    for (int q = sub; q < np; q += R) acc = fmaf(e_s[q], r[((long long)b * HW + p0 + q) * C + c], acc);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  float* out = part + ((long long)b * nblk + blk) * (C + 2);
  if (threadIdx.x < C) {
    float s = 0.f;
    for (int q = 0; q < R; ++q) s += red[q * C + threadIdx.x];
    out[threadIdx.x] = s;
  }
  if (threadIdx.x == 0) { out[C] = m; out[C + 1] = esum; }
}

// Stage 2 (one block per batch item): combine partials, context -> 1x1 -> LeakyReLU(0.2) -> 1x1
__global__ void gc_stage2_kernel(const float* part, int nblk, int C, const float* w1, const float* w2, float* add) {
  extern __shared__ float sm[];  // ctx[C], hid[C], scale[nblk]
  float* ctx = sm;
  float* hid = sm + C;
  float* scale = sm + 2 * C;
  __shared__ float red[256];
  const int b = blockIdx.x;
  const float* pb = part + (long long)b * nblk * (C + 2);
  // global max of the per-block maxima (tree reduction, fixed order)
  float gm = -INFINITY;
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) gm = fmaxf(gm, pb[(long long)k * (C + 2) + C]);
  red[threadIdx.x] = gm;
  __syncthreads();
  for (int st = blockDim.x / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  const float gmax_s = red[0];
  __syncthreads();
  float den = 0.f;
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) {
    const float sc = expf(pb[(long long)k * (C + 2) + C] - gmax_s);
    scale[k] = sc;
    den = fmaf(pb[(long long)k * (C + 2) + C + 1], sc, den);
  }
  red[threadIdx.x] = den;
  __syncthreads();
  for (int st = blockDim.x / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  const float denom_s = red[0];
  __syncthreads();
  // ctx[c]: 256 threads = (256/C) partial sums per channel over the blocks, then a fixed-order combine
  {
    const int R = blockDim.x / C;
    const int sub = threadIdx.x / C, c = threadIdx.x % C;
    float s = 0.f;
    if (sub < R)
      for (int k = sub; k < nblk; k += R) s = fmaf(pb[(long long)k * (C + 2) + c], scale[k], s);
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
      float t = 0.f;
      for (int q = 0; q < R; ++q) t += red[q * C + threadIdx.x];
      ctx[threadIdx.x] = t / denom_s;
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(w1[o * C + c], ctx[c], s);
    hid[o] = s >= 0.f ? s : 0.2f * s;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(w2[o * C + c], hid[c], s);
    add[(long long)b * C + o] = s;
  }
}

// ---- 16-bit activation storage (trunk16 mode): 4 channels = 8 bytes ---------------------------------------------------
template <int DT>
__device__ __forceinline__ float4 ld4(const void* base, long long quad) {
  if (DT == FCVSR_F32) return reinterpret_cast<const float4*>(base)[quad];
  const uint2 v = reinterpret_cast<const uint2*>(base)[quad];
  if (DT == FCVSR_BF16)
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 a = __builtin_bit_cast(h2, v.x), b = __builtin_bit_cast(h2, v.y);
  return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}

template <int DT>
__device__ __forceinline__ void st4(void* base, long long quad, float4 x) {
  if (DT == FCVSR_F32) {
    reinterpret_cast<float4*>(base)[quad] = x;
  } else if (DT == FCVSR_BF16) {
    typedef __attribute__((ext_vector_type(4))) __bf16 b4;
    const b4 c = {(__bf16)x.x, (__bf16)x.y, (__bf16)x.z, (__bf16)x.w};
    reinterpret_cast<uint2*>(base)[quad] = __builtin_bit_cast(uint2, c);
  } else {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 c = {(_Float16)x.x, (_Float16)x.y, (_Float16)x.z, (_Float16)x.w};
    reinterpret_cast<uint2*>(base)[quad] = __builtin_bit_cast(uint2, c);
  }
}

template <int DT>
__global__ void gc_apply_kernel(const float4* r, const float* add, const void* z, void* out, float slope,
                                long long HWCq, int Cq, long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int b = (int)(t / HWCq);
  const int cq = (int)(t % Cq);
  const float* a = add + (long long)b * Cq * 4 + cq * 4;
  const float4 rr = r[t], zz = ld4<DT>(z, t);
  float4 v = make_float4(rr.x + a[0], rr.y + a[1], rr.z + a[2], rr.w + a[3]);
  v.x = v.x >= 0.f ? v.x : v.x * slope; v.y = v.y >= 0.f ? v.y : v.y * slope;
  v.z = v.z >= 0.f ? v.z : v.z * slope; v.w = v.w >= 0.f ? v.w : v.w * slope;
  st4<DT>(out, t, make_float4(v.x + zz.x, v.y + zz.y, v.z + zz.z, v.w + zz.w));
}

__device__ __forceinline__ float4 f4_axpy(float a, float4 x, float4 y) {
  return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), fmaf(a, x.w, y.w));
}

// out = x + rs*r + mean2x2(dn) + bilinear_x2(up)   (F.interpolate align_corners=False semantics)
template <int DT>
__global__ void xscale_kernel(const void* x, const void* r, float rs, const void* dn, const void* up, void* out,
                              int B, int H, int W, int Cq) {
  const long long total = (long long)B * H * W * Cq;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int cq = (int)(t % Cq);
  const long long pg = t / Cq;
  const int xx = (int)(pg % W);
  const int yy = (int)((pg / W) % H);
  const int b = (int)(pg / ((long long)W * H));
  float4 acc = f4_axpy(rs, ld4<DT>(r, t), ld4<DT>(x, t));
  if (dn) {
    const int H2 = 2 * H, W2 = 2 * W;
    const long long d0 = ((long long)b * H2 * W2) * Cq + cq;
    const float4 a00 = ld4<DT>(dn, d0 + ((long long)(2 * yy) * W2 + 2 * xx) * Cq);
    const float4 a01 = ld4<DT>(dn, d0 + ((long long)(2 * yy) * W2 + 2 * xx + 1) * Cq);
    const float4 a10 = ld4<DT>(dn, d0 + ((long long)(2 * yy + 1) * W2 + 2 * xx) * Cq);
    const float4 a11 = ld4<DT>(dn, d0 + ((long long)(2 * yy + 1) * W2 + 2 * xx + 1) * Cq);
    // torch upsample_bilinear2d order: lerp along x inside each row, then along y (weights 0.5)
    const float4 top = make_float4(0.5f * a00.x + 0.5f * a01.x, 0.5f * a00.y + 0.5f * a01.y, 0.5f * a00.z + 0.5f * a01.z,
                                   0.5f * a00.w + 0.5f * a01.w);
    const float4 bot = make_float4(0.5f * a10.x + 0.5f * a11.x, 0.5f * a10.y + 0.5f * a11.y, 0.5f * a10.z + 0.5f * a11.z,
                                   0.5f * a10.w + 0.5f * a11.w);
    acc = f4_axpy(0.5f, top, acc);
    acc = f4_axpy(0.5f, bot, acc);
  }
  if (up) {
    const int Hh = H / 2, Wh = W / 2;
    float sy = 0.5f * ((float)yy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = 0.5f * ((float)xx + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Hh - 1 ? 1 : 0), x1 = x0 + (x0 < Wh - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const long long u0 = ((long long)b * Hh * Wh) * Cq + cq;
    const float4 u00 = ld4<DT>(up, u0 + ((long long)y0 * Wh + x0) * Cq), u01 = ld4<DT>(up, u0 + ((long long)y0 * Wh + x1) * Cq);
    const float4 u10 = ld4<DT>(up, u0 + ((long long)y1 * Wh + x0) * Cq), u11 = ld4<DT>(up, u0 + ((long long)y1 * Wh + x1) * Cq);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    acc = f4_axpy(w00, u00, acc);
    acc = f4_axpy(w01, u01, acc);
    acc = f4_axpy(w10, u10, acc);
    acc = f4_axpy(w11, u11, acc);
  }
  st4<DT>(out, t, acc);
}


// ---- level-grouped variants: up to 3 pyramid levels per launch ----------------------------------------------------------
struct GcFinishLv { const float* part; float* add; int nparts; };
struct GcFinishArgs { GcFinishLv lv[3]; };

__global__ void gc_stage2_levels_kernel(GcFinishArgs a, int C, const float* w1, const float* w2) {
  // same arithmetic as gc_stage2_kernel; blockIdx.y selects the level
  const GcFinishLv L = a.lv[blockIdx.y];
  extern __shared__ float sm[];  // ctx[C], hid[C], scale[nparts]
  float* ctx = sm;
  float* hid = sm + C;
  float* scale = sm + 2 * C;
  __shared__ float red[256];
  __shared__ __align__(16) float red4_s[4 * 256];
  const int b = blockIdx.x, nblk = L.nparts;
  const float* pb = L.part + (long long)b * nblk * (C + 2);
  float gm = -INFINITY;
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) gm = fmaxf(gm, pb[(long long)k * (C + 2) + C]);
  red[threadIdx.x] = gm;
  __syncthreads();
  for (int st = blockDim.x / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  const float gmax_s = red[0];
  __syncthreads();
  float den = 0.f;
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) {
    const float sc = expf(pb[(long long)k * (C + 2) + C] - gmax_s);
    scale[k] = sc;
    den = fmaf(pb[(long long)k * (C + 2) + C + 1], sc, den);
  }
  red[threadIdx.x] = den;
  __syncthreads();
  for (int st = blockDim.x / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  const float denom_s = red[0];
  __syncthreads();
  if (C % 4 == 0 && blockDim.x % (C / 4) == 0 && (C + 2) % 2 == 0) {
    // ctx[c]: thread = (sub-lane, 4-channel group): 256 / (C/4) partial sums per channel over the tiles with 8-byte loads
    // (rows are C+2 floats long, so a 4-channel group is 8-byte aligned), then a fixed-order combine
    const int Cq = C / 4, R = blockDim.x / Cq;
    const int sub = threadIdx.x / Cq, cq = threadIdx.x % Cq;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight tiles' loads in flight per step, added in the same (ascending k) order as the rolled loop: identical sums.
    // (rolled, every iteration waited for its own 16 bytes: ~225 dependent round trips for the full-resolution level)
    int k = sub;
    for (; k + 7 * R < nblk; k += 8 * R) {
      float2 lo[8], hi[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float2* pr = reinterpret_cast<const float2*>(pb + (long long)(k + u * R) * (C + 2) + cq * 4);
        lo[u] = pr[0]; hi[u] = pr[1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float sc = scale[k + u * R];
        acc.x = fmaf(lo[u].x, sc, acc.x); acc.y = fmaf(lo[u].y, sc, acc.y); acc.z = fmaf(hi[u].x, sc, acc.z); acc.w = fmaf(hi[u].y, sc, acc.w);
      }
    }
    for (; k < nblk; k += R) {
      const float2* pr = reinterpret_cast<const float2*>(pb + (long long)k * (C + 2) + cq * 4);
      const float2 lo = pr[0], hi = pr[1];
      const float sc = scale[k];
      acc.x = fmaf(lo.x, sc, acc.x); acc.y = fmaf(lo.y, sc, acc.y); acc.z = fmaf(hi.x, sc, acc.z); acc.w = fmaf(hi.y, sc, acc.w);
    }
    float4* red4 = reinterpret_cast<float4*>(red4_s);
    red4[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < Cq) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q = 0; q < R; ++q) {
        const float4 v = red4[q * Cq + threadIdx.x];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      ctx[threadIdx.x * 4] = t.x / denom_s; ctx[threadIdx.x * 4 + 1] = t.y / denom_s;
      ctx[threadIdx.x * 4 + 2] = t.z / denom_s; ctx[threadIdx.x * 4 + 3] = t.w / denom_s;
    }
  } else
  {
    const int R = blockDim.x / C;
    const int sub = threadIdx.x / C, c = threadIdx.x % C;
    float s = 0.f;
    if (sub < R)
      for (int k = sub; k < nblk; k += R) s = fmaf(pb[(long long)k * (C + 2) + c], scale[k], s);
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
      float t = 0.f;
      for (int q = 0; q < R; ++q) t += red[q * C + threadIdx.x];
      ctx[threadIdx.x] = t / denom_s;
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(w1[o * C + c], ctx[c], s);
    hid[o] = s >= 0.f ? s : 0.2f * s;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(w2[o * C + c], hid[c], s);
    L.add[(long long)b * C + o] = s;
  }
}

struct GcApplyLv {
  const void* r;
  const float* add;
  const void* z;
  void* out;
  void* pool;
  int B, H, W;
  int blk_begin;           // first workgroup of this level
};
struct GcApplyArgs { GcApplyLv lv[3]; int n; };

template <int DT, bool R16>
__device__ __forceinline__ float4 gc_apply_one(const void* r, const float* a, const void* z, void* out, long long q, float slope) {
  const float4 rr = R16 ? ld4<DT>(r, q) : ld4<FCVSR_F32>(r, q), zz = ld4<DT>(z, q);
  float4 v = make_float4(rr.x + a[0], rr.y + a[1], rr.z + a[2], rr.w + a[3]);
  v.x = v.x >= 0.f ? v.x : v.x * slope; v.y = v.y >= 0.f ? v.y : v.y * slope;
  v.z = v.z >= 0.f ? v.z : v.z * slope; v.w = v.w >= 0.f ? v.w : v.w * slope;
  v = make_float4(v.x + zz.x, v.y + zz.y, v.z + zz.z, v.w + zz.w);
  st4<DT>(out, q, v);
  return v;
}

// value as it is stored (rounded to the io dtype): the pooled tensor averages what a reader of `out` would see
template <int DT>
__device__ __forceinline__ float4 as_stored(float4 v) {
  if (DT == FCVSR_F32) return v;
  if (DT == FCVSR_BF16) {
    typedef __attribute__((ext_vector_type(4))) __bf16 b4;
    const b4 c = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    return make_float4((float)c[0], (float)c[1], (float)c[2], (float)c[3]);
  }
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  const h4 c = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  return make_float4((float)c[0], (float)c[1], (float)c[2], (float)c[3]);
}

template <int DT, bool R16>
__global__ void gc_apply_levels_kernel(GcApplyArgs a, float slope, int Cq) {
  int li = 0;
  if (a.n > 1 && (int)blockIdx.x >= a.lv[1].blk_begin) li = 1;
  if (a.n > 2 && (int)blockIdx.x >= a.lv[2].blk_begin) li = 2;
  const GcApplyLv L = a.lv[li];
  const long long t = (long long)(blockIdx.x - L.blk_begin) * blockDim.x + threadIdx.x;
  if (L.pool) {
    // thread = (2x2 pixel block, channel quad): four outputs and their average
    const int H2 = L.H >> 1, W2 = L.W >> 1;
    const long long total = (long long)L.B * H2 * W2 * Cq;
    if (t >= total) return;
    const int cq = (int)(t % Cq);
    const long long pg = t / Cq;
    const int x2 = (int)(pg % W2), y2 = (int)((pg / W2) % H2), b = (int)(pg / ((long long)W2 * H2));
    const float* ad = L.add + (long long)b * Cq * 4 + cq * 4;
    const long long q00 = (((long long)b * L.H + 2 * y2) * L.W + 2 * x2) * Cq + cq;
    const float4 v00 = as_stored<DT>(gc_apply_one<DT, R16>(L.r, ad, L.z, L.out, q00, slope));
    const float4 v01 = as_stored<DT>(gc_apply_one<DT, R16>(L.r, ad, L.z, L.out, q00 + Cq, slope));
    const float4 v10 = as_stored<DT>(gc_apply_one<DT, R16>(L.r, ad, L.z, L.out, q00 + (long long)L.W * Cq, slope));
    const float4 v11 = as_stored<DT>(gc_apply_one<DT, R16>(L.r, ad, L.z, L.out, q00 + (long long)L.W * Cq + Cq, slope));
    // torch upsample_bilinear2d order: lerp along x inside each row, then along y (weights 0.5)
    const float4 top = make_float4(0.5f * v00.x + 0.5f * v01.x, 0.5f * v00.y + 0.5f * v01.y, 0.5f * v00.z + 0.5f * v01.z, 0.5f * v00.w + 0.5f * v01.w);
    const float4 bot = make_float4(0.5f * v10.x + 0.5f * v11.x, 0.5f * v10.y + 0.5f * v11.y, 0.5f * v10.z + 0.5f * v11.z, 0.5f * v10.w + 0.5f * v11.w);
    st4<DT>(L.pool, t, make_float4(0.5f * top.x + 0.5f * bot.x, 0.5f * top.y + 0.5f * bot.y, 0.5f * top.z + 0.5f * bot.z, 0.5f * top.w + 0.5f * bot.w));
  } else {
    const long long HWCq = (long long)L.H * L.W * Cq;
    if (t >= HWCq * L.B) return;
    const int b = (int)(t / HWCq), cq = (int)(t % Cq);
    gc_apply_one<DT, R16>(L.r, L.add + (long long)b * Cq * 4 + cq * 4, L.z, L.out, t, slope);
  }
}

struct XsLv {
  const void* x; const void* r; const void* dn; const void* up; void* out;
  float rs;
  int dn_pooled, B, H, W, blk_begin;
};
struct XsArgs { XsLv lv[3]; int n; };

// V quads (4 V channels) per thread: V = 2 in the 16-bit modes makes every access 16 bytes per lane
template <int V> struct QuadV { float4 p[V]; };

template <int DT, int V>
__device__ __forceinline__ QuadV<V> ldq(const void* base, long long idx) {       // idx counts groups of V quads
  QuadV<V> r;
  if (V == 2 && DT != FCVSR_F32) {
    const uint4 v = reinterpret_cast<const uint4*>(base)[idx];
    if (DT == FCVSR_BF16) {
      r.p[0] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
      r.p[V - 1] = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16),
                               __uint_as_float(v.w & 0xffff0000u));
    } else {
      typedef __attribute__((ext_vector_type(2))) _Float16 h2;
      const h2 a = __builtin_bit_cast(h2, v.x), b = __builtin_bit_cast(h2, v.y), c = __builtin_bit_cast(h2, v.z),
               d = __builtin_bit_cast(h2, v.w);
      r.p[0] = make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
      r.p[V - 1] = make_float4((float)c[0], (float)c[1], (float)d[0], (float)d[1]);
    }
    return r;
  }
#pragma unroll
  for (int i = 0; i < V; ++i) r.p[i] = ld4<DT>(base, idx * V + i);
  return r;
}

template <int DT, int V>
__device__ __forceinline__ void stq(void* base, long long idx, const QuadV<V>& x) {
  if (V == 2 && DT != FCVSR_F32) {
    uint4 o;
    if (DT == FCVSR_BF16) {
      typedef __attribute__((ext_vector_type(4))) __bf16 b4;
      const b4 c0 = {(__bf16)x.p[0].x, (__bf16)x.p[0].y, (__bf16)x.p[0].z, (__bf16)x.p[0].w};
      const b4 c1 = {(__bf16)x.p[V - 1].x, (__bf16)x.p[V - 1].y, (__bf16)x.p[V - 1].z, (__bf16)x.p[V - 1].w};
      const uint2 u0 = __builtin_bit_cast(uint2, c0), u1 = __builtin_bit_cast(uint2, c1);
      o = make_uint4(u0.x, u0.y, u1.x, u1.y);
    } else {
      typedef __attribute__((ext_vector_type(4))) _Float16 h4;
      const h4 c0 = {(_Float16)x.p[0].x, (_Float16)x.p[0].y, (_Float16)x.p[0].z, (_Float16)x.p[0].w};
      const h4 c1 = {(_Float16)x.p[V - 1].x, (_Float16)x.p[V - 1].y, (_Float16)x.p[V - 1].z, (_Float16)x.p[V - 1].w};
      const uint2 u0 = __builtin_bit_cast(uint2, c0), u1 = __builtin_bit_cast(uint2, c1);
      o = make_uint4(u0.x, u0.y, u1.x, u1.y);
    }
    reinterpret_cast<uint4*>(base)[idx] = o;
    return;
  }
#pragma unroll
  for (int i = 0; i < V; ++i) st4<DT>(base, idx * V + i, x.p[i]);
}

template <int V>
__device__ __forceinline__ QuadV<V> qv_axpy(float a, const QuadV<V>& x, const QuadV<V>& y) {
  QuadV<V> r;
#pragma unroll
  for (int i = 0; i < V; ++i) r.p[i] = f4_axpy(a, x.p[i], y.p[i]);
  return r;
}
template <int V>
__device__ __forceinline__ QuadV<V> qv_half_sum(const QuadV<V>& a, const QuadV<V>& b) {       // 0.5 a + 0.5 b, torch's order
  QuadV<V> r;
#pragma unroll
  for (int i = 0; i < V; ++i)
    r.p[i] = make_float4(0.5f * a.p[i].x + 0.5f * b.p[i].x, 0.5f * a.p[i].y + 0.5f * b.p[i].y,
                         0.5f * a.p[i].z + 0.5f * b.p[i].z, 0.5f * a.p[i].w + 0.5f * b.p[i].w);
  return r;
}

// Cq counts the per-pixel groups of V quads (C / (4 V))
template <int DT, int V>
__global__ void xscale_levels_kernel(XsArgs a, int Cq) {
  int li = 0;
  if (a.n > 1 && (int)blockIdx.x >= a.lv[1].blk_begin) li = 1;
  if (a.n > 2 && (int)blockIdx.x >= a.lv[2].blk_begin) li = 2;
  const XsLv L = a.lv[li];
  const int H = L.H, W = L.W;
  const long long total = (long long)L.B * H * W * Cq;
  const long long t = (long long)(blockIdx.x - L.blk_begin) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int cq = (int)(t % Cq);
  const long long pg = t / Cq;
  const int xx = (int)(pg % W);
  const int yy = (int)((pg / W) % H);
  const int b = (int)(pg / ((long long)W * H));
  QuadV<V> acc = qv_axpy<V>(L.rs, ldq<DT, V>(L.r, t), ldq<DT, V>(L.x, t));
  if (L.dn) {
    if (L.dn_pooled) {
      const QuadV<V> d = ldq<DT, V>(L.dn, t);
#pragma unroll
      for (int i = 0; i < V; ++i)
        acc.p[i] = make_float4(acc.p[i].x + d.p[i].x, acc.p[i].y + d.p[i].y, acc.p[i].z + d.p[i].z, acc.p[i].w + d.p[i].w);
    } else {
      const int H2 = 2 * H, W2 = 2 * W;
      const long long d0 = ((long long)b * H2 * W2) * Cq + cq;
      const QuadV<V> a00 = ldq<DT, V>(L.dn, d0 + ((long long)(2 * yy) * W2 + 2 * xx) * Cq);
      const QuadV<V> a01 = ldq<DT, V>(L.dn, d0 + ((long long)(2 * yy) * W2 + 2 * xx + 1) * Cq);
      const QuadV<V> a10 = ldq<DT, V>(L.dn, d0 + ((long long)(2 * yy + 1) * W2 + 2 * xx) * Cq);
      const QuadV<V> a11 = ldq<DT, V>(L.dn, d0 + ((long long)(2 * yy + 1) * W2 + 2 * xx + 1) * Cq);
      const QuadV<V> top = qv_half_sum<V>(a00, a01), bot = qv_half_sum<V>(a10, a11);
      acc = qv_axpy<V>(0.5f, top, acc);
      acc = qv_axpy<V>(0.5f, bot, acc);
    }
  }
  if (L.up) {
    const int Hh = H / 2, Wh = W / 2;
    float sy = 0.5f * ((float)yy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = 0.5f * ((float)xx + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Hh - 1 ? 1 : 0), x1 = x0 + (x0 < Wh - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const long long u0 = ((long long)b * Hh * Wh) * Cq + cq;
    const QuadV<V> u00 = ldq<DT, V>(L.up, u0 + ((long long)y0 * Wh + x0) * Cq), u01 = ldq<DT, V>(L.up, u0 + ((long long)y0 * Wh + x1) * Cq);
    const QuadV<V> u10 = ldq<DT, V>(L.up, u0 + ((long long)y1 * Wh + x0) * Cq), u11 = ldq<DT, V>(L.up, u0 + ((long long)y1 * Wh + x1) * Cq);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    acc = qv_axpy<V>(w00, u00, acc);
    acc = qv_axpy<V>(w01, u01, acc);
    acc = qv_axpy<V>(w10, u10, acc);
    acc = qv_axpy<V>(w11, u11, acc);
  }
  stq<DT, V>(L.out, t, acc);
}


// ---------------------------------------------------------------------------------------------------------------------
// Level 0 of BlockRCB's second half in one pass (16-bit storage modes).  The generic sequence writes R0 = lrelu(r + add) + z
// (gc_apply_levels) and reads it back for out0 = x + 2 R0 + up2(up.0(R1)) (xscale_levels); at the full-resolution level that
// round trip is the largest elementwise cost of the block.  R0 is needed by nobody else: the down path only wants its 2x2
// average.  Thread = (2x2 pixel block, 8 channels): it forms the four R0 values (rounded to the storage type exactly where
// the two-kernel path stores them, so results are bit-identical), writes their average P0 and the four finished outputs.
// The bilinear x2 up-sample (align_corners=False) of a 2x2 block touches the 3x3 neighbourhood of the half-resolution source.
struct Rcb0Args {
  const void* x; const void* r; const float* add; const void* z; const void* up; void* out; void* pool;
  float slope, rs;
  int B, H, W;
};

template <int DT>
__global__ void __launch_bounds__(256) rcb_level0_kernel(Rcb0Args a, int Cq) {      // Cq = C / 8
  const int H2 = a.H >> 1, W2 = a.W >> 1;
  const long long total = (long long)a.B * H2 * W2 * Cq;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int cq = (int)(t % Cq);
  const long long pg = t / Cq;
  const int x2 = (int)(pg % W2), y2 = (int)((pg / W2) % H2), b = (int)(pg / ((long long)W2 * H2));
  const float* ad = a.add + (long long)b * Cq * 8 + cq * 8;
  const float4 ad0 = *reinterpret_cast<const float4*>(ad), ad1 = *reinterpret_cast<const float4*>(ad + 4);
  const long long q00 = (((long long)b * a.H + 2 * y2) * a.W + 2 * x2) * Cq + cq;
  const long long qs[4] = {q00, q00 + Cq, q00 + (long long)a.W * Cq, q00 + (long long)a.W * Cq + Cq};
  QuadV<2> rr[4], zz[4], xx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { rr[i] = ldq<DT, 2>(a.r, qs[i]); zz[i] = ldq<DT, 2>(a.z, qs[i]); xx[i] = ldq<DT, 2>(a.x, qs[i]); }
  // 3 x 3 neighbourhood of the half-resolution up source, indices clamped like upsample_bilinear2d's y1 / x1
  const int ys[3] = {y2 > 0 ? y2 - 1 : 0, y2, y2 < H2 - 1 ? y2 + 1 : y2};
  const int xs[3] = {x2 > 0 ? x2 - 1 : 0, x2, x2 < W2 - 1 ? x2 + 1 : x2};
  QuadV<2> U[3][3];
  const long long u0 = ((long long)b * H2 * W2) * Cq + cq;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) U[i][j] = ldq<DT, 2>(a.up, u0 + ((long long)ys[i] * W2 + xs[j]) * Cq);
  QuadV<2> R[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4 av = h ? ad1 : ad0, rv = rr[i].p[h], zv = zz[i].p[h];
      float4 v = make_float4(rv.x + av.x, rv.y + av.y, rv.z + av.z, rv.w + av.w);
      v.x = v.x >= 0.f ? v.x : v.x * a.slope; v.y = v.y >= 0.f ? v.y : v.y * a.slope;
      v.z = v.z >= 0.f ? v.z : v.z * a.slope; v.w = v.w >= 0.f ? v.w : v.w * a.slope;
      R[i].p[h] = as_stored<DT>(make_float4(v.x + zv.x, v.y + zv.y, v.z + zv.z, v.w + zv.w));
    }
  }
  {
    const QuadV<2> top = qv_half_sum<2>(R[0], R[1]), bot = qv_half_sum<2>(R[2], R[3]);
    stq<DT, 2>(a.pool, t, qv_half_sum<2>(top, bot));
  }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    // even output row 2 y2: source y = y2 - 0.25 (clamped at 0); odd row: y2 + 0.25
    const float ly = dy ? 0.25f : (y2 > 0 ? 0.75f : 0.f);
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const float lx = dx ? 0.25f : (x2 > 0 ? 0.75f : 0.f);
      const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
      QuadV<2> acc = qv_axpy<2>(a.rs, R[dy * 2 + dx], xx[dy * 2 + dx]);
      acc = qv_axpy<2>(w00, U[dy][dx], acc);
      acc = qv_axpy<2>(w01, U[dy][dx + 1], acc);
      acc = qv_axpy<2>(w10, U[dy + 1][dx], acc);
      acc = qv_axpy<2>(w11, U[dy + 1][dx + 1], acc);
      stq<DT, 2>(a.out, qs[dy * 2 + dx], acc);
    }
  }
}

}  // namespace fcvsr

using namespace fcvsr;

static bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

extern "C" int fcvsr_gc_context(const float* r, const float* wmask, const float* w1, const float* w2, int B, int H, int W,
                                int C, float* add, float* scratch, int64_t scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(r && wmask && w1 && w2 && add && scratch, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && C >= 4 && C <= 256 && C % 4 == 0, "C in 4..256, C%4==0");
  FCVSR_CHECK_ARG(al16(r) && al16(wmask), "16-byte alignment");
  const long long HW = (long long)H * W;
  const int nblk = cdiv(HW, kGcPix);
  FCVSR_CHECK_ARG(scratch_elems >= (long long)B * nblk * (C + 2), "scratch too small");
  FCVSR_CHECK_ARG((2ll * C + nblk) * 4 <= 60 * 1024, "image too large for stage-2 LDS");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gc_stage1_kernel, dim3(nblk, B), dim3(kGcPix), 0, st, r, wmask, HW, C, scratch);
  FCVSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(gc_stage2_kernel, dim3(B), dim3(256), (2 * C + nblk) * sizeof(float), st, (const float*)scratch, nblk,
                     C, w1, w2, add);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_gc_finish(const float* partial, int nparts, const float* w1, const float* w2, int B, int C, float* add,
                               void* stream) {
  FCVSR_CHECK_ARG(partial && w1 && w2 && add, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && nparts > 0 && C >= 4 && C <= 256, "bad sizes");
  FCVSR_CHECK_ARG((2ll * C + nparts) * 4 <= 60 * 1024, "too many partials for stage-2 LDS");
  hipLaunchKernelGGL(gc_stage2_kernel, dim3(B), dim3(256), (2 * C + nparts) * sizeof(float), (hipStream_t)stream, partial,
                     nparts, C, w1, w2, add);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_gc_apply(const float* r, const float* add, const void* z, void* out, int io_dtype, float slope, int B,
                              int H, int W, int C, void* stream) {
  FCVSR_CHECK_ARG(r && add && z && out, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "C%4==0 required");
  FCVSR_CHECK_ARG(al16(r) && al16(z) && al16(out), "16-byte alignment");
  const int Cq = C / 4;
  const long long HWCq = (long long)H * W * Cq;
  const long long total = HWCq * B;
  FCVSR_CHECK_ARG(io_dtype == FCVSR_F32 || io_dtype == FCVSR_BF16 || io_dtype == FCVSR_F16, "bad io_dtype");
  if (io_dtype == FCVSR_F32)
    hipLaunchKernelGGL((gc_apply_kernel<FCVSR_F32>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)r,
                       add, z, out, slope, HWCq, Cq, total);
  else if (io_dtype == FCVSR_BF16)
    hipLaunchKernelGGL((gc_apply_kernel<FCVSR_BF16>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)r,
                       add, z, out, slope, HWCq, Cq, total);
  else
    hipLaunchKernelGGL((gc_apply_kernel<FCVSR_F16>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)r,
                       add, z, out, slope, HWCq, Cq, total);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_xscale(const void* x, const void* r, float r_scale, const void* dn, const void* up, void* out,
                            int io_dtype, int B, int H, int W, int C, void* stream) {
  FCVSR_CHECK_ARG(x && r && out, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "C%4==0 required");
  FCVSR_CHECK_ARG(!up || (H % 2 == 0 && W % 2 == 0), "up source needs even H,W");
  FCVSR_CHECK_ARG(al16(x) && al16(r) && al16(out) && al16(dn) && al16(up), "16-byte alignment");
  const long long total = (long long)B * H * W * (C / 4);
  FCVSR_CHECK_ARG(io_dtype == FCVSR_F32 || io_dtype == FCVSR_BF16 || io_dtype == FCVSR_F16, "bad io_dtype");
  if (io_dtype == FCVSR_F32)
    hipLaunchKernelGGL((xscale_kernel<FCVSR_F32>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, r, r_scale, dn,
                       up, out, B, H, W, C / 4);
  else if (io_dtype == FCVSR_BF16)
    hipLaunchKernelGGL((xscale_kernel<FCVSR_BF16>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, r, r_scale, dn,
                       up, out, B, H, W, C / 4);
  else
    hipLaunchKernelGGL((xscale_kernel<FCVSR_F16>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, r, r_scale, dn,
                       up, out, B, H, W, C / 4);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_rcb_level0(const void* x, const void* r, const float* add, const void* z, const void* up, void* out,
                                void* pool, float slope, float r_scale, int io_dtype, int B, int H, int W, int C, void* stream) {
  FCVSR_CHECK_ARG(x && r && add && z && up && out && pool, "null pointer");
  FCVSR_CHECK_ARG(io_dtype == FCVSR_BF16 || io_dtype == FCVSR_F16, "16-bit storage modes only");
  FCVSR_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "even H, W required");
  FCVSR_CHECK_ARG(C > 0 && C % 8 == 0, "C%8==0 required");
  FCVSR_CHECK_ARG(al16(x) && al16(r) && al16(z) && al16(up) && al16(out) && al16(pool) && al16(add), "16-byte alignment");
  FCVSR_CHECK_ARG(out != x && out != r && out != z && out != up && pool != r && pool != z && pool != x && pool != up,
                  "outputs must not alias inputs (neighbouring threads read the same up pixels)");
  Rcb0Args a{x, r, add, z, up, out, pool, slope, r_scale, B, H, W};
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / 8);
  if (io_dtype == FCVSR_BF16)
    hipLaunchKernelGGL((rcb_level0_kernel<FCVSR_BF16>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a, C / 8);
  else
    hipLaunchKernelGGL((rcb_level0_kernel<FCVSR_F16>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a, C / 8);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_gc_finish_levels(const fcvsr_gc_finish_level* lv, int n_levels, const float* w1, const float* w2, int B,
                                      int C, void* stream) {
  FCVSR_CHECK_ARG(lv && w1 && w2 && n_levels >= 1 && n_levels <= 3, "1..3 levels");
  FCVSR_CHECK_ARG(B > 0 && C >= 4 && C <= 256, "bad sizes");
  GcFinishArgs a;
  int maxp = 0;
  for (int l = 0; l < 3; ++l) {
    const fcvsr_gc_finish_level& s = lv[l < n_levels ? l : 0];
    FCVSR_CHECK_ARG(s.partial && s.add && s.nparts > 0, "null level");
    a.lv[l].part = s.partial; a.lv[l].add = s.add; a.lv[l].nparts = s.nparts;
    maxp = s.nparts > maxp ? s.nparts : maxp;
  }
  FCVSR_CHECK_ARG((2ll * C + maxp) * 4 <= 60 * 1024, "too many partials for stage-2 LDS");
  hipLaunchKernelGGL(gc_stage2_levels_kernel, dim3(B, n_levels), dim3(256), (2 * C + maxp) * sizeof(float), (hipStream_t)stream,
                     a, C, w1, w2);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_gc_apply_levels(const fcvsr_gc_apply_level* lv, int n_levels, int io_dtype, int r_dtype, float slope,
                                     int C, void* stream) {
  FCVSR_CHECK_ARG(lv && n_levels >= 1 && n_levels <= 3, "1..3 levels");
  FCVSR_CHECK_ARG(C > 0 && C % 4 == 0, "C%4==0 required");
  FCVSR_CHECK_ARG(io_dtype == FCVSR_F32 || io_dtype == FCVSR_BF16 || io_dtype == FCVSR_F16, "bad io_dtype");
  GcApplyArgs a;
  a.n = n_levels;
  int blocks = 0;
  for (int l = 0; l < 3; ++l) {
    const fcvsr_gc_apply_level& s = lv[l < n_levels ? l : 0];
    FCVSR_CHECK_ARG(s.r && s.add && s.z && s.out && s.B > 0 && s.H > 0 && s.W > 0, "null level");
    FCVSR_CHECK_ARG(al16(s.r) && al16(s.z) && al16(s.out) && al16(s.pool), "16-byte alignment");
    FCVSR_CHECK_ARG(!s.pool || (s.H % 2 == 0 && s.W % 2 == 0), "pooled output needs even H, W");
    GcApplyLv& d = a.lv[l];
    d.r = s.r; d.add = s.add; d.z = s.z; d.out = s.out; d.pool = s.pool; d.B = s.B; d.H = s.H; d.W = s.W;
    d.blk_begin = blocks;
    if (l < n_levels) {
      const long long items = s.pool ? (long long)s.B * (s.H / 2) * (s.W / 2) * (C / 4) : (long long)s.B * s.H * s.W * (C / 4);
      blocks += cdiv(items, 256);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  FCVSR_CHECK_ARG(r_dtype == FCVSR_F32 || r_dtype == io_dtype, "r is f32 or stored like z/out");
  const bool r16 = r_dtype != FCVSR_F32;
#define FCVSR_GAL(DTV, R16V) hipLaunchKernelGGL((gc_apply_levels_kernel<DTV, R16V>), dim3(blocks), dim3(256), 0, st, a, slope, C / 4)
  if (io_dtype == FCVSR_F32) FCVSR_GAL(FCVSR_F32, false);
  else if (io_dtype == FCVSR_BF16) { if (r16) FCVSR_GAL(FCVSR_BF16, true); else FCVSR_GAL(FCVSR_BF16, false); }
  else { if (r16) FCVSR_GAL(FCVSR_F16, true); else FCVSR_GAL(FCVSR_F16, false); }
#undef FCVSR_GAL
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_xscale_levels(const fcvsr_xscale_level* lv, int n_levels, int io_dtype, int C, void* stream) {
  FCVSR_CHECK_ARG(lv && n_levels >= 1 && n_levels <= 3, "1..3 levels");
  FCVSR_CHECK_ARG(C > 0 && C % 4 == 0, "C%4==0 required");
  FCVSR_CHECK_ARG(io_dtype == FCVSR_F32 || io_dtype == FCVSR_BF16 || io_dtype == FCVSR_F16, "bad io_dtype");
  XsArgs a;
  a.n = n_levels;
  int blocks = 0;
  const int V = (io_dtype != FCVSR_F32 && C % 8 == 0) ? 2 : 1;         // 16-byte accesses in the 16-bit modes
  for (int l = 0; l < 3; ++l) {
    const fcvsr_xscale_level& s = lv[l < n_levels ? l : 0];
    FCVSR_CHECK_ARG(s.x && s.r && s.out && s.B > 0 && s.H > 0 && s.W > 0, "null level");
    FCVSR_CHECK_ARG(!s.up || (s.H % 2 == 0 && s.W % 2 == 0), "up source needs even H,W");
    FCVSR_CHECK_ARG(al16(s.x) && al16(s.r) && al16(s.out) && al16(s.dn) && al16(s.up), "16-byte alignment");
    XsLv& d = a.lv[l];
    d.x = s.x; d.r = s.r; d.dn = s.dn; d.up = s.up; d.out = s.out; d.rs = s.r_scale; d.dn_pooled = s.dn_pooled;
    d.B = s.B; d.H = s.H; d.W = s.W; d.blk_begin = blocks;
    if (l < n_levels) blocks += cdiv((long long)s.B * s.H * s.W * (C / (4 * V)), 256);
  }
  hipStream_t st = (hipStream_t)stream;
#define FCVSR_XSL(DTV, VV) hipLaunchKernelGGL((xscale_levels_kernel<DTV, VV>), dim3(blocks), dim3(256), 0, st, a, C / (4 * VV))
  if (io_dtype == FCVSR_F32) FCVSR_XSL(FCVSR_F32, 1);
  else if (io_dtype == FCVSR_BF16) { if (V == 2) FCVSR_XSL(FCVSR_BF16, 2); else FCVSR_XSL(FCVSR_BF16, 1); }
  else { if (V == 2) FCVSR_XSL(FCVSR_F16, 2); else FCVSR_XSL(FCVSR_F16, 1); }
#undef FCVSR_XSL
  FCVSR_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================================
// ContextBlock softmax-pool partials from a STORED 16-bit r (all pyramid levels in one launch).  The lean 3x3 kernel can emit
// these from its epilogue (fcvsr_conv_desc.gc_wmask), but that epilogue costs it a third of its speed (185 vs 127 us for the
// 64 -> 64 layer); with the resident-weight kernel taking the layer (102 us) it is cheaper to re-read r once (203 MB per
// launch set, mostly from L2 / Infinity Cache right behind the producer) and compute the partials here.
// One workgroup = one 4 x 32 pixel tile = the partition (and [C+2] record layout) of the fused epilogue, so
// fcvsr_gc_finish_levels consumes either.  Two passes over registers: tile maximum of the logits first, then exp / sums, all in
// fixed order (bit-reproducible).
#include "mfma_util.h"
namespace fcvsr {

struct GcPartLv { const uint16_t* r; float* partial; int B, H, W, tiles_x, tiles_y, tile_begin; };
struct GcPartArgs { GcPartLv lv[3]; int n_levels; const float* wmask; };

template <bool BF16>
__global__ __launch_bounds__(256) void gc_partial16_levels_kernel(GcPartArgs a) {
  __shared__ float red_m[4];
  __shared__ float red_s[4];
  __shared__ __align__(16) float red_a[4][64];
  int t = blockIdx.x, li = 0;
  if (a.n_levels > 1 && t >= a.lv[1].tile_begin) li = 1;
  if (a.n_levels > 2 && t >= a.lv[2].tile_begin) li = 2;
  const GcPartLv L = a.lv[li];
  t -= L.tile_begin;
  const int per_img = L.tiles_x * L.tiles_y;
  const int b = t / per_img, t2 = t - b * per_img;
  const int ty0 = (t2 / L.tiles_x) * 4, tx0 = (t2 % L.tiles_x) * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int oct = tid & 7, ps = tid >> 3;                // 8 channels per lane, 32 pixels per pass, 4 passes (= 4 tile rows)
  const float4 w0 = *reinterpret_cast<const float4*>(a.wmask + oct * 8), w1 = *reinterpret_cast<const float4*>(a.wmask + oct * 8 + 4);
  float v[4][8], logit[4];
  bool ok[4];
  uint4 rawv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {                            // the four loads first, unconditional (clamped): predicated, each
    const int py = ty0 + j, px = tx0 + ps;                 // load + conversion was its own block and its own round trip
    ok[j] = py < L.H && px < L.W;
    const int cy = py < L.H ? py : L.H - 1, cx = px < L.W ? px : L.W - 1;
    rawv[j] = *reinterpret_cast<const uint4*>(L.r + (((long long)b * L.H + cy) * L.W + cx) * 64 + oct * 8);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint4 raw = ok[j] ? rawv[j] : make_uint4(0, 0, 0, 0);
    cvt16x4_to_f32<BF16>(make_uint2(raw.x, raw.y), v[j]);
    cvt16x4_to_f32<BF16>(make_uint2(raw.z, raw.w), v[j] + 4);
    float p = v[j][0] * w0.x + v[j][1] * w0.y + v[j][2] * w0.z + v[j][3] * w0.w + v[j][4] * w1.x + v[j][5] * w1.y + v[j][6] * w1.z + v[j][7] * w1.w;
    p += __shfl_xor(p, 1); p += __shfl_xor(p, 2); p += __shfl_xor(p, 4);      // the 8 lanes of a pixel
    logit[j] = ok[j] ? p : -INFINITY;
  }
  float m = fmaxf(fmaxf(logit[0], logit[1]), fmaxf(logit[2], logit[3]));
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red_m[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));                // the tile has at least one live pixel
  float s = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float e = ok[j] ? expf(logit[j] - m) : 0.f;
    s += e;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = fmaf(e, v[j][c], acc[c]);
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {                      // the 8 pixel slots of a wave (same channel octet)
    s += __shfl_xor(s, o);
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] += __shfl_xor(acc[c], o);
  }
  if (lane < 8) {
    *reinterpret_cast<float4*>(&red_a[wave][lane * 8]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(&red_a[wave][lane * 8 + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    if (lane == 0) red_s[wave] = s;
  }
  __syncthreads();
  if (tid < 64) {
    float* pp = L.partial + ((long long)b * per_img + t2) * 66;
    pp[tid] = (red_a[0][tid] + red_a[1][tid]) + (red_a[2][tid] + red_a[3][tid]);
    if (tid == 0) { pp[64] = m; pp[65] = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]); }
  }
}

}  // namespace fcvsr

extern "C" int fcvsr_gc_partial_levels(const fcvsr_gc_partial_level* lv, int n_levels, int r_dtype, const float* wmask, int C, void* stream) {
  FCVSR_CHECK_ARG(lv && n_levels >= 1 && n_levels <= 3 && wmask && C == 64, "1..3 levels, 64 channels");
  FCVSR_CHECK_ARG(r_dtype == FCVSR_BF16 || r_dtype == FCVSR_F16, "r must be stored in 16 bit");
  FCVSR_CHECK_ARG(((uintptr_t)wmask % 16) == 0, "wmask must be 16-byte aligned");
  fcvsr::GcPartArgs a;
  a.n_levels = n_levels; a.wmask = wmask;
  int tiles = 0;
  for (int l = 0; l < 3; ++l) {
    const fcvsr_gc_partial_level& s = lv[l < n_levels ? l : 0];
    FCVSR_CHECK_ARG(s.r && s.partial && s.B > 0 && s.H > 0 && s.W > 0 && ((uintptr_t)s.r % 16) == 0, "bad level");
    fcvsr::GcPartLv& L = a.lv[l];
    L.r = (const uint16_t*)s.r; L.partial = s.partial; L.B = s.B; L.H = s.H; L.W = s.W;
    L.tiles_x = fcvsr::cdiv(s.W, 32); L.tiles_y = fcvsr::cdiv(s.H, 4);
    L.tile_begin = tiles;
    if (l < n_levels) tiles += s.B * L.tiles_x * L.tiles_y;
  }
  if (r_dtype == FCVSR_BF16) hipLaunchKernelGGL(fcvsr::gc_partial16_levels_kernel<true>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(fcvsr::gc_partial16_levels_kernel<false>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
