// MGAAbk elementwise / gather kernels (reference CVSR_freq.py:1365-1547): CorrBlock lookup, channel sums + CALayer gate,
// ConvBlk tail, flow_warp, separable adaptive conv (SAC) passes.  All HBM-bound: one lane = one pixel x 4 channels
// (16-byte accesses), consecutive lanes = consecutive channel quads of the same pixel, then consecutive pixels.
#include "common.h"
#include "reduce.h"

namespace fcvsr {

// ---- CorrBlock (:1279-1337) ------------------------------------------------------------------------------------------
// corr[c=i*n+j][y][x] = I_p[y+j-r][x+i-r] (zero outside the C/2 x 2 image), I_p = the C consecutive floats at flat
// offset p*C of the NCHW-contiguous product buffer P = x1f*x2f/sqrt(C)  (raw .view() reinterpretation in the reference).
__global__ void corr_lookup_kernel(const float* x1f, const float* x2f, long long ps, int B, int H, int Wf, int C,
                                   int radius, int xw, View dst, float norm_div) {
  const int n = 2 * radius + 1;
  const int nn = n * n;
  const long long total = (long long)B * H * xw * dst.c;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % dst.c);
  const long long pixg = t / dst.c;
  const int x = (int)(pixg % xw);
  const int y = (int)((pixg / xw) % H);
  const int b = (int)(pixg / ((long long)xw * H));
  float v = 0.f;
  if (c < nn) {
    const int i = c / n, j = c % n;
    const int col = x + i - radius, row = y + j - radius;
    if (col >= 0 && col <= 1 && row >= 0 && row < C / 2) {
      const long long HW = (long long)H * Wf;
      const long long e = ((long long)y * Wf + x) * C + row * 2 + col;  // flat index inside batch item (NCHW order)
      const int ch = (int)(e / HW);
      const long long pp = e % HW;
      const long long src = ((long long)b * HW + pp) * ps + ch;         // NHWC address of (ch, pp)
      v = (x1f[src] * x2f[src]) / norm_div;
    }
  }
  dst.p[(long long)b * dst.sb + (long long)y * dst.sy + (long long)x * dst.sx + (long long)c * dst.sc] = v;
}

// ---- channel sums ------------------------------------------------------------------------------------------------------
struct ViewSumF {
  View v;
  int W;
  __device__ void operator()(int b, long long p, int c, float* out) const {
    const int x = (int)(p % W);
    const long long y = p / W;
    out[0] = v.p[(long long)b * v.sb + y * v.sy + (long long)x * v.sx + (long long)c * v.sc];
  }
};

// ---- CALayer gate (:1812-1828) ---------------------------------------------------------------------------------------
__global__ void ca_gate_kernel(const float* sum, float inv_hw, const float* w1, const float* w2, int C, int CR,
                               float* gate) {
  extern __shared__ float sm[];  // mean[C], hid[CR]
  float* mean = sm;
  float* hid = sm + C;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) mean[c] = sum[(long long)b * C + c] * inv_hw;
  __syncthreads();
  for (int h = threadIdx.x; h < CR; h += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(w1[h * C + c], mean[c], s);
    hid[h] = fmaxf(s, 0.f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int h = 0; h < CR; ++h) s = fmaf(w2[c * CR + h], hid[h], s);
    gate[(long long)b * C + c] = 1.f / (1.f + expf(-s));
  }
}

// ---- ConvBlk tail (:355-356) * sim, split into (real, imag) planes (:1495-1498) -----------------------------------------
__global__ void convblk_tail_kernel(const float4* u, const float* gate, const float4* sim, int B, int ndir, long long HW,
                                    float* spec, long long ps, int re_off, int im_off, int g_stride, int g0) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)ndir * B * HW) return;
  const long long pix = t % HW;
  const int bn = (int)(t / HW);
  const int dir = bn / B, b = bn % B;
  const float4 uu = u[t];
  const float4 ss = sim[(long long)b * HW + pix];
  const float* g = gate + bn * 4;
  const float o0 = fmaf(uu.x, g[0], uu.x) * ss.x;
  const float o1 = fmaf(uu.y, g[1], uu.y) * ss.y;
  const float o2 = fmaf(uu.z, g[2], uu.z) * ss.z;
  const float o3 = fmaf(uu.w, g[3], uu.w) * ss.w;
  float* px = spec + ((long long)b * HW + pix) * ps;
  const int gi = (g0 + dir * g_stride) * 2;
  px[re_off + gi] = o0;
  px[re_off + gi + 1] = o1;
  px[im_off + gi] = o2;
  px[im_off + gi + 1] = o3;
}

// ---- flow_warp (:1188-1227): bilinear, zeros padding ---------------------------------------------------------------------
__global__ void warp_kernel(View src, View off, int B, int H, int W, View dst) {
  const int CQ = src.c / 4;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * H * W * CQ) return;
  const int cq = (int)(t % CQ);
  const long long pixg = t / CQ;
  const int x = (int)(pixg % W);
  const int y = (int)((pixg / W) % H);
  const int b = (int)(pixg / ((long long)W * H));
  const float* op = off.p + (long long)b * off.sb + (long long)y * off.sy + (long long)x * off.sx;
  const float fx = (float)x + op[0];
  const float fy = (float)y + op[off.sc];
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float wx1 = fx - x0f, wy1 = fy - y0f;
  const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  // guard against non-finite / huge offsets before the int conversion
  const bool sane = (fx > -2.f) && (fx < (float)W + 1.f) && (fy > -2.f) && (fy < (float)H + 1.f);
  const int x0 = sane ? (int)x0f : -4, y0 = sane ? (int)y0f : -4;
  const float* sp = src.p + (long long)b * src.sb + cq * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int xi = x0 + dx, yi = y0 + dy;
      if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
        const float w = (dy ? wy1 : wy0) * (dx ? wx1 : wx0);
        const float4 v = *reinterpret_cast<const float4*>(sp + (long long)yi * src.sy + (long long)xi * src.sx);
        acc.x = fmaf(v.x, w, acc.x); acc.y = fmaf(v.y, w, acc.y);
        acc.z = fmaf(v.z, w, acc.z); acc.w = fmaf(v.w, w, acc.w);
      }
    }
  }
  *reinterpret_cast<float4*>(dst.p + (long long)b * dst.sb + (long long)y * dst.sy + (long long)x * dst.sx + cq * 4) = acc;
}

// ---- SAC passes (:1253-1276).  k1 channel index = c*3+t; a lane's 4 channels own 12 consecutive kernel floats ---------
template <bool HORIZ>
__global__ void sac_kernel(View s, View k1, View fin, float slope, int B, int H, int W, View dst) {
  const int CQ = s.c / 4;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * H * W * CQ) return;
  const int cq = (int)(t % CQ);
  const long long pixg = t / CQ;
  const int x = (int)(pixg % W);
  const int y = (int)((pixg / W) % H);
  const int b = (int)(pixg / ((long long)W * H));
  const float* kp = k1.p + (long long)b * k1.sb + (long long)y * k1.sy + (long long)x * k1.sx + cq * 12;
  const float4 ka = *reinterpret_cast<const float4*>(kp);
  const float4 kb = *reinterpret_cast<const float4*>(kp + 4);
  const float4 kc = *reinterpret_cast<const float4*>(kp + 8);
  const float kk[4][3] = {{ka.x, ka.y, ka.z}, {ka.w, kb.x, kb.y}, {kb.z, kb.w, kc.x}, {kc.y, kc.z, kc.w}};
  const float* sp = s.p + (long long)b * s.sb + cq * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int tt = 0; tt < 3; ++tt) {
    int yy = y, xx = x;
    if (HORIZ) { xx = x + tt - 1; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx); }
    else { yy = y + tt - 1; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy); }
    const float4 v = *reinterpret_cast<const float4*>(sp + (long long)yy * s.sy + (long long)xx * s.sx);
    acc.x = fmaf(v.x, kk[0][tt], acc.x); acc.y = fmaf(v.y, kk[1][tt], acc.y);
    acc.z = fmaf(v.z, kk[2][tt], acc.z); acc.w = fmaf(v.w, kk[3][tt], acc.w);
  }
  if (HORIZ) {
    const float4 f = *reinterpret_cast<const float4*>(fin.p + (long long)b * fin.sb + (long long)y * fin.sy +
                                                      (long long)x * fin.sx + cq * 4);
    acc.x += f.x; acc.y += f.y; acc.z += f.z; acc.w += f.w;
    acc.x = acc.x >= 0.f ? acc.x : acc.x * slope; acc.y = acc.y >= 0.f ? acc.y : acc.y * slope;
    acc.z = acc.z >= 0.f ? acc.z : acc.z * slope; acc.w = acc.w >= 0.f ? acc.w : acc.w * slope;
  }
  *reinterpret_cast<float4*>(dst.p + (long long)b * dst.sb + (long long)y * dst.sy + (long long)x * dst.sx + cq * 4) = acc;
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_corr_lookup(const float* x1f, const float* x2f, int64_t pix_stride, int B, int H, int Wf, int C,
                                 int radius, int x_count, const fcvsr_view* dst, void* stream) {
  FCVSR_CHECK_ARG(x1f && x2f && dst && dst->ptr, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && Wf > 0 && C > 0 && C % 2 == 0 && radius >= 0 && pix_stride >= C, "bad sizes");
  FCVSR_CHECK_ARG(dst->c >= (2 * radius + 1) * (2 * radius + 1) && dst->dtype == FCVSR_F32, "dst: f32, >= (2r+1)^2 channels");
  FCVSR_CHECK_ARG(x_count > 0 && x_count <= Wf, "x_count must be in [1, Wf]");
  const long long total = (long long)B * H * x_count * dst->c;
  hipLaunchKernelGGL(corr_lookup_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x1f, x2f,
                     (long long)pix_stride, B, H, Wf, C, radius, x_count, to_view(*dst), sqrtf((float)C));
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_channel_sum(const fcvsr_view* src, int B, int H, int W, float* out, float* scratch,
                                 int64_t scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(src && src->ptr && out && scratch, "null pointer");
  FCVSR_CHECK_ARG(src->c >= 1 && src->c <= 256, "1..256 channels");
  const long long npix = (long long)H * W;
  const int nblk = red_blocks(npix);
  FCVSR_CHECK_ARG(scratch_elems >= (long long)B * nblk * src->c, "scratch too small");
  ViewSumF f{to_view(*src), W};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((reduce_stage1<1, ViewSumF>), dim3(nblk, B), dim3(kRedThreads), 0, st, f, B, npix, src->c, scratch);
  FCVSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_stage2, dim3(B), dim3(kRedThreads), 0, st, (const float*)scratch, B, nblk, src->c, out);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_ca_gate(const float* sum, float inv_hw, const float* w1, const float* w2, int B, int c, int cr,
                             float* gate, void* stream) {
  FCVSR_CHECK_ARG(sum && w1 && w2 && gate, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && c > 0 && cr > 0 && c <= 4096 && cr <= 4096, "bad sizes");
  hipLaunchKernelGGL(ca_gate_kernel, dim3(B), dim3(64), (c + cr) * sizeof(float), (hipStream_t)stream, sum, inv_hw, w1,
                     w2, c, cr, gate);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_convblk_tail(const float* u, const float* gate, const float* sim, int B, int ndir, int H, int Wf,
                                  float* spec, int64_t pix_stride, int re_off, int im_off, int g_stride, int g0,
                                  void* stream) {
  FCVSR_CHECK_ARG(u && gate && sim && spec, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && ndir > 0 && H > 0 && Wf > 0, "bad sizes");
  FCVSR_CHECK_ARG(((uintptr_t)u % 16 == 0) && ((uintptr_t)sim % 16 == 0), "u/sim must be 16-byte aligned");
  const long long HW = (long long)H * Wf;
  const long long total = (long long)ndir * B * HW;
  hipLaunchKernelGGL(convblk_tail_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)u,
                     gate, (const float4*)sim, B, ndir, HW, spec, (long long)pix_stride, re_off, im_off, g_stride, g0);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

static bool quad_ok(const fcvsr_view* v) { return v && v->ptr && v->dtype == FCVSR_F32 && vec4_ok(*v); }

extern "C" int fcvsr_warp(const fcvsr_view* src, const fcvsr_view* off, int B, int H, int W, const fcvsr_view* dst,
                          void* stream) {
  FCVSR_CHECK_ARG(quad_ok(src) && quad_ok(dst), "src/dst must be f32, channel-contiguous, 16-byte aligned, c%4==0");
  FCVSR_CHECK_ARG(off && off->ptr && off->c >= 2 && off->dtype == FCVSR_F32, "off needs 2 channels");
  FCVSR_CHECK_ARG(src->c == dst->c, "channel mismatch");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0, "bad sizes");
  const long long total = (long long)B * H * W * (src->c / 4);
  hipLaunchKernelGGL(warp_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, to_view(*src), to_view(*off), B,
                     H, W, to_view(*dst));
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_sac_v(const fcvsr_view* s, const fcvsr_view* k1, int B, int H, int W, const fcvsr_view* dst,
                           void* stream) {
  FCVSR_CHECK_ARG(quad_ok(s) && quad_ok(dst) && quad_ok(k1), "views must be f32, channel-contiguous, aligned");
  FCVSR_CHECK_ARG(k1->c == 3 * s->c && s->c == dst->c, "k1 must have 3*C channels");
  const long long total = (long long)B * H * W * (s->c / 4);
  hipLaunchKernelGGL((sac_kernel<false>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, to_view(*s),
                     to_view(*k1), to_view(*s), 0.f, B, H, W, to_view(*dst));
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_sac_h(const fcvsr_view* v, const fcvsr_view* k1, const fcvsr_view* feat_in, float slope, int B, int H,
                           int W, const fcvsr_view* dst, void* stream) {
  FCVSR_CHECK_ARG(quad_ok(v) && quad_ok(dst) && quad_ok(k1) && quad_ok(feat_in), "views must be f32, contiguous, aligned");
  FCVSR_CHECK_ARG(k1->c == 3 * v->c && v->c == dst->c && feat_in->c == v->c, "channel mismatch");
  const long long total = (long long)B * H * W * (v->c / 4);
  hipLaunchKernelGGL((sac_kernel<true>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, to_view(*v),
                     to_view(*k1), to_view(*feat_in), slope, B, H, W, to_view(*dst));
  FCVSR_LAUNCH_CHECK();
  return 0;
}
