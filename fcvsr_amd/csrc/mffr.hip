// MultiFreq_Refinment elementwise kernels (reference CVSR_freq.py:2104-2133 DivEnh, :2201-2254).
// The band split itself is fcvsr_rfft2 + masked fcvsr_irfft2 (fft.hip).  Everything here is HBM-bound streaming over dense
// NHWC (B,H,W,C) f32 tensors; reductions are deterministic two-stage sums (reduce.h).
#include "common.h"
#include "reduce.h"

namespace fcvsr {

struct DivEnhExprF {
  const float* f;
  const float* s_f;
  const float* s_o;
  const float* a;
  const float* b;
  const float* mean_sum;  // [B][C] sums of f (first block only)
  float inv_hw;
  long long HW;
  int C;
  int first;
  __device__ void eval(int bi, long long p, int c, float& e1, float& e2, float& fv) const {
    const long long i = ((long long)bi * HW + p) * C + c;
    fv = f[i];
    const float aa = 0.2f * a[c], bb = b[c];
    if (first) {
      const float t = fv - mean_sum[bi * C + c] * inv_hw;
      e1 = aa * t * fv + bb * fv;
      e2 = 0.f;
    } else {
      const float so = s_o[i];
      const float t = fv - s_f[i] + 0.2f * so;
      e1 = aa * t * fv + bb * fv;
      e2 = aa * so * fv + bb * fv;
    }
  }
  __device__ void operator()(int bi, long long p, int c, float* out) const {
    float e1, e2, fv;
    eval(bi, p, c, e1, e2, fv);
    out[0] = e1;
    out[1] = e2;
  }
};

__global__ void divenh_apply_kernel(DivEnhExprF ex, float* s_f, float* s_o, const float* g1, const float* g2, int B) {
  const long long total = (long long)B * ex.HW * ex.C;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % ex.C);
  const long long pg = t / ex.C;
  const long long p = pg % ex.HW;
  const int bi = (int)(pg / ex.HW);
  float e1, e2, fv;
  ex.eval(bi, p, c, e1, e2, fv);
  float o = e1 * g1[bi * ex.C + c];
  if (!ex.first) o += e2 * g2[bi * ex.C + c];
  if (ex.first) { s_f[t] = fv; s_o[t] = o; }
  else { s_f[t] += fv; s_o[t] += o; }
}

// apply step of band i fused with the reduction that the NEXT step needs (same partition and order as reduce_stage1 on
// DivEnhExprF, so the sums are bit-identical to the two-kernel sequence): the running sums s_f, s_o are read and written
// once instead of being read again by a separate reduction pass.
struct DivEnhApplyNextF {
  DivEnhExprF ex;          // band i (f, s_f, s_o, a, b, mean, first)
  float* s_f;
  float* s_o;
  const float* g1;
  const float* g2;
  const float* f_next;     // band i+1 (next_kind 0)
  const float* a_next;
  const float* b_next;
  int next_kind;           // 0: e1, e2 of band i+1;  1: (new s_o, 0) = channel sums for the final CALayer
  __device__ void operator()(int bi, long long p, int c, float* out) const {
    const long long i = ((long long)bi * ex.HW + p) * ex.C + c;
    float e1, e2, fv;
    ex.eval(bi, p, c, e1, e2, fv);
    float o = e1 * g1[bi * ex.C + c];
    if (!ex.first) o += e2 * g2[bi * ex.C + c];
    float nf, no;
    if (ex.first) { nf = fv; no = o; }
    else { nf = s_f[i] + fv; no = s_o[i] + o; }
    s_f[i] = nf;
    s_o[i] = no;
    if (next_kind == 0) {
      const float fn = f_next[i];
      const float aa = 0.2f * a_next[c], bb = b_next[c];
      const float t = fn - nf + 0.2f * no;
      out[0] = aa * t * fn + bb * fn;
      out[1] = aa * no * fn + bb * fn;
    } else {
      out[0] = no;
      out[1] = 0.f;
    }
  }
};

// The same step with 16-byte accesses (C % 4 == 0): thread = (pixel sub-lane, 4-channel group), kRedPix pixels per block,
// fixed summation order (per thread over its pixels, then over the sub-lanes) - deterministic like reduce_stage1.
__global__ __launch_bounds__(kRedThreads) void divenh_apply_next_v4_kernel(DivEnhApplyNextF e, int B, float* partial) {
  __shared__ float4 sm[2][kRedThreads];
  const int C = e.ex.C, Cq = C >> 2;
  const long long npix = e.ex.HW;
  const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int R = kRedThreads / Cq;
  const int sub = threadIdx.x / Cq, cq = threadIdx.x % Cq;
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
  if (sub < R) {
    const int c = cq * 4;
    const float4 av = *reinterpret_cast<const float4*>(e.ex.a + c), bv = *reinterpret_cast<const float4*>(e.ex.b + c);
    const float aa[4] = {0.2f * av.x, 0.2f * av.y, 0.2f * av.z, 0.2f * av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
    float g1v[4], g2v[4] = {0.f, 0.f, 0.f, 0.f}, mean[4] = {0.f, 0.f, 0.f, 0.f}, an[4] = {0.f, 0.f, 0.f, 0.f}, bn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g1v[k] = e.g1[b * C + c + k];
      if (!e.ex.first) g2v[k] = e.g2[b * C + c + k];
      if (e.ex.first) mean[k] = e.ex.mean_sum[b * C + c + k] * e.ex.inv_hw;
      if (e.next_kind == 0) { an[k] = 0.2f * e.a_next[c + k]; bn[k] = e.b_next[c + k]; }
    }
    const long long p0 = (long long)blk * kRedPix;
    const long long p1 = (p0 + kRedPix < npix) ? p0 + kRedPix : npix;
    for (long long p = p0 + sub; p < p1; p += R) {
      const long long i = ((long long)b * npix + p) * C + c;
      const float4 f4 = *reinterpret_cast<const float4*>(e.ex.f + i);
      float4 sf4 = make_float4(0.f, 0.f, 0.f, 0.f), so4 = sf4, fn4 = sf4;
      if (!e.ex.first) { sf4 = *reinterpret_cast<const float4*>(e.s_f + i); so4 = *reinterpret_cast<const float4*>(e.s_o + i); }
      if (e.next_kind == 0) fn4 = *reinterpret_cast<const float4*>(e.f_next + i);
      const float fv[4] = {f4.x, f4.y, f4.z, f4.w}, sf[4] = {sf4.x, sf4.y, sf4.z, sf4.w}, so[4] = {so4.x, so4.y, so4.z, so4.w};
      const float fn[4] = {fn4.x, fn4.y, fn4.z, fn4.w};
      float nf[4], no[4], o1[4], o2[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float e1, e2;                                   // exactly DivEnhExprF::eval
        if (e.ex.first) {
          const float t = fv[k] - mean[k];
          e1 = aa[k] * t * fv[k] + bb[k] * fv[k];
          e2 = 0.f;
        } else {
          const float t = fv[k] - sf[k] + 0.2f * so[k];
          e1 = aa[k] * t * fv[k] + bb[k] * fv[k];
          e2 = aa[k] * so[k] * fv[k] + bb[k] * fv[k];
        }
        float o = e1 * g1v[k];
        if (!e.ex.first) o += e2 * g2v[k];
        if (e.ex.first) { nf[k] = fv[k]; no[k] = o; }
        else { nf[k] = sf[k] + fv[k]; no[k] = so[k] + o; }
        if (e.next_kind == 0) {
          const float t = fn[k] - nf[k] + 0.2f * no[k];
          o1[k] = an[k] * t * fn[k] + bn[k] * fn[k];
          o2[k] = an[k] * no[k] * fn[k] + bn[k] * fn[k];
        } else {
          o1[k] = no[k];
          o2[k] = 0.f;
        }
      }
      *reinterpret_cast<float4*>(e.s_f + i) = make_float4(nf[0], nf[1], nf[2], nf[3]);
      *reinterpret_cast<float4*>(e.s_o + i) = make_float4(no[0], no[1], no[2], no[3]);
      a1.x += o1[0]; a1.y += o1[1]; a1.z += o1[2]; a1.w += o1[3];
      a2.x += o2[0]; a2.y += o2[1]; a2.z += o2[2]; a2.w += o2[3];
    }
  }
  sm[0][threadIdx.x] = a1;
  sm[1][threadIdx.x] = a2;
  __syncthreads();
  if (threadIdx.x < Cq) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < R; ++r) {
        const float4 v = sm[k][r * Cq + threadIdx.x];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      *reinterpret_cast<float4*>(partial + (((long long)k * B + b) * nblk + blk) * C + threadIdx.x * 4) = t;
    }
  }
}

template <int XD>
__device__ __forceinline__ float4 ld_x4(const void* base, long long quad) {
  if (XD == FCVSR_F32) return reinterpret_cast<const float4*>(base)[quad];
  const uint2 v = reinterpret_cast<const uint2*>(base)[quad];
  if (XD == FCVSR_BF16)
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 a = __builtin_bit_cast(h2, v.x), b = __builtin_bit_cast(h2, v.y);
  return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}

template <int DT, int XD>
__global__ void scale_add_kernel(const float4* z, const float* gate, const void* x, void* out, long long HWCq, int Cq,
                                 long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int b = (int)(t / HWCq);
  const int cq = (int)(t % Cq);
  const float* g = gate + (long long)b * Cq * 4 + cq * 4;
  const float4 zz = z[t], xx = ld_x4<XD>(x, t);
  const float4 o = make_float4(fmaf(zz.x, g[0], xx.x), fmaf(zz.y, g[1], xx.y), fmaf(zz.z, g[2], xx.z), fmaf(zz.w, g[3], xx.w));
  if (DT == FCVSR_F32) {
    reinterpret_cast<float4*>(out)[t] = o;
  } else if (DT == FCVSR_BF16) {
    typedef __attribute__((ext_vector_type(4))) __bf16 b4;
    const b4 c = {(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
    reinterpret_cast<uint2*>(out)[t] = __builtin_bit_cast(uint2, c);
  } else {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 c = {(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
    reinterpret_cast<uint2*>(out)[t] = __builtin_bit_cast(uint2, c);
  }
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_divenh(int mode, int first, const float* f, float* s_f, float* s_o, const float* a, const float* b,
                            const float* mean_f_sum, float inv_hw, const float* g1, const float* g2, float* sums,
                            float* scratch, int64_t scratch_elems, int B, int H, int W, int C, void* stream) {
  FCVSR_CHECK_ARG(f && s_f && s_o && a && b, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C <= 256, "bad sizes");
  FCVSR_CHECK_ARG(!first || mean_f_sum, "first block needs the sums of f");
  DivEnhExprF ex{f, s_f, s_o, a, b, mean_f_sum, inv_hw, (long long)H * W, C, first};
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    FCVSR_CHECK_ARG(sums && scratch, "reduce mode needs sums and scratch");
    const int nblk = red_blocks(ex.HW);
    FCVSR_CHECK_ARG(scratch_elems >= 2ll * B * nblk * C, "scratch too small");
    hipLaunchKernelGGL((reduce_stage1<2, DivEnhExprF>), dim3(nblk, B), dim3(kRedThreads), 0, st, ex, B, ex.HW, C, scratch);
    FCVSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_stage2, dim3(2 * B), dim3(kRedThreads), 0, st, (const float*)scratch, 2 * B, nblk, C, sums);
    FCVSR_LAUNCH_CHECK();
  } else {
    FCVSR_CHECK_ARG(g1 && (first || g2), "apply mode needs gates");
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(divenh_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, ex, s_f, s_o, g1, g2, B);
    FCVSR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int fcvsr_divenh_apply_next(int first, const float* f, float* s_f, float* s_o, const float* a, const float* b,
                                       const float* mean_f_sum, float inv_hw, const float* g1, const float* g2,
                                       const float* f_next, const float* a_next, const float* b_next, float* sums,
                                       float* scratch, int64_t scratch_elems, int B, int H, int W, int C, void* stream) {
  FCVSR_CHECK_ARG(f && s_f && s_o && a && b && g1 && (first || g2) && sums && scratch, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C <= 256, "bad sizes");
  FCVSR_CHECK_ARG(!first || mean_f_sum, "first block needs the sums of f");
  FCVSR_CHECK_ARG((f_next == nullptr) == (a_next == nullptr) && (f_next == nullptr) == (b_next == nullptr), "next band: all or none");
  DivEnhApplyNextF ex{{f, s_f, s_o, a, b, mean_f_sum, inv_hw, (long long)H * W, C, first}, s_f, s_o, g1, g2, f_next, a_next,
                      b_next, f_next ? 0 : 1};
  const long long HW = (long long)H * W;
  const int nblk = red_blocks(HW);
  FCVSR_CHECK_ARG(scratch_elems >= 2ll * B * nblk * C, "scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const bool v4 = C % 4 == 0 && kRedThreads % (C / 4) == 0 && ((uintptr_t)f % 16) == 0 && ((uintptr_t)s_f % 16) == 0 &&
                  ((uintptr_t)s_o % 16) == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 &&
                  ((uintptr_t)scratch % 16) == 0 && (f_next == nullptr || ((uintptr_t)f_next % 16) == 0);
  if (v4) hipLaunchKernelGGL(divenh_apply_next_v4_kernel, dim3(nblk, B), dim3(kRedThreads), 0, st, ex, B, scratch);
  else hipLaunchKernelGGL((reduce_stage1<2, DivEnhApplyNextF>), dim3(nblk, B), dim3(kRedThreads), 0, st, ex, B, HW, C, scratch);
  FCVSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_stage2, dim3(2 * B), dim3(kRedThreads), 0, st, (const float*)scratch, 2 * B, nblk, C, sums);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_scale_add(const float* z, const float* gate, const void* x, int x_dtype, void* out, int out_dtype, int B,
                               int H, int W, int C, void* stream) {
  FCVSR_CHECK_ARG(z && gate && x && out, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "C%4==0 required");
  FCVSR_CHECK_ARG(((uintptr_t)z % 16 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0), "16-byte alignment");
  const int Cq = C / 4;
  const long long HWCq = (long long)H * W * Cq;
  const long long total = HWCq * B;
  FCVSR_CHECK_ARG(out_dtype == FCVSR_F32 || out_dtype == FCVSR_BF16 || out_dtype == FCVSR_F16, "bad out_dtype");
  FCVSR_CHECK_ARG(x_dtype == FCVSR_F32 || x_dtype == out_dtype, "x must be f32 or stored like out");
  dim3 grid(cdiv(total, 256));
  hipStream_t st = (hipStream_t)stream;
#define FCVSR_SA(OD, XD) hipLaunchKernelGGL((scale_add_kernel<OD, XD>), grid, dim3(256), 0, st, (const float4*)z, gate, x, out, HWCq, Cq, total)
  if (out_dtype == FCVSR_F32) FCVSR_SA(FCVSR_F32, FCVSR_F32);
  else if (out_dtype == FCVSR_BF16) { if (x_dtype == FCVSR_F32) FCVSR_SA(FCVSR_BF16, FCVSR_F32); else FCVSR_SA(FCVSR_BF16, FCVSR_BF16); }
  else { if (x_dtype == FCVSR_F32) FCVSR_SA(FCVSR_F16, FCVSR_F32); else FCVSR_SA(FCVSR_F16, FCVSR_F16); }
#undef FCVSR_SA
  FCVSR_LAUNCH_CHECK();
  return 0;
}
