// Training path: backward of one IAC iteration (reference CVSR_freq.py:1230-1250: flow_warp :1188-1227, SAC :1253-1276)
//     s = flow_warp(prev, off);  v[y][x] = sum_t s[clamp(y+t-1)][x] K[c*3+t][y][x];  h[y][x] = sum_t v[y][clamp(x+t-1)] K[c*3+t][y][x];
//     out = LeakyReLU_slope(h + feat_in)
// as two launches on NHWC f32 tensors (the forward is fcvsr_warp / fcvsr_sac_v / fcvsr_sac_h, which leave s and v in memory):
//   fcvsr_iac_bwd_sac : g_out -> g_feat_in (+=), g_v, g_K (both passes' contributions, assign or +=)
//   fcvsr_iac_bwd_warp: g_v -> g_s (vertical pass transposed) -> g_prev (bilinear scatter, float atomics into a zeroed tensor) and g_off
// Under autograd these were ~150 torch kernels per iteration and direction (gathers, strided multiplies, pads, masks, slices).
#include "common.h"

namespace fcvsr {

__device__ __forceinline__ void ld12(const float* p, float k[4][3]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4), c = *reinterpret_cast<const float4*>(p + 8);
  k[0][0] = a.x; k[0][1] = a.y; k[0][2] = a.z; k[1][0] = a.w; k[1][1] = b.x; k[1][2] = b.y;
  k[2][0] = b.z; k[2][1] = b.w; k[2][2] = c.x; k[3][0] = c.y; k[3][1] = c.z; k[3][2] = c.w;
}

// thread = (pixel, 4 consecutive channels).  gy, yout, v, s, gfin, gv: dense (B,H,W,C); k1 / gk: views with 3*C channels.
__global__ __launch_bounds__(256) void iac_bwd_sac_kernel(const float* __restrict__ gy, const float* __restrict__ yout, const float* __restrict__ v,
                                                          const float* __restrict__ s, View k1, float slope, int B, int H, int W, int C,
                                                          float* __restrict__ gfin, int fin_accumulate, float* __restrict__ gv, View gk, int k_accumulate) {
  const int CQ = C / 4;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * H * W * CQ) return;
  const int cq = (int)(t % CQ);
  const long long pixg = t / CQ;
  const int x = (int)(pixg % W);
  const int y = (int)((pixg / W) % H);
  const int b = (int)(pixg / ((long long)W * H));
  const long long rowb = ((long long)b * H + y) * W;
  const int xl = x > 0 ? x - 1 : 0, xr = x < W - 1 ? x + 1 : W - 1;
  float gh[3][4];                                   // gradient at the pre-activation h at x-1, x, x+1 (clamped positions)
  const int xs[3] = {xl, x, xr};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 g4 = *reinterpret_cast<const float4*>(gy + (rowb + xs[i]) * C + cq * 4);
    const float4 y4 = *reinterpret_cast<const float4*>(yout + (rowb + xs[i]) * C + cq * 4);
    gh[i][0] = y4.x > 0.f ? g4.x : g4.x * slope; gh[i][1] = y4.y > 0.f ? g4.y : g4.y * slope;
    gh[i][2] = y4.z > 0.f ? g4.z : g4.z * slope; gh[i][3] = y4.w > 0.f ? g4.w : g4.w * slope;
  }
  float kL[4][3], kC[4][3], kR[4][3];
  const float* kb = k1.p + (long long)b * k1.sb + (long long)y * k1.sy + cq * 12;
  ld12(kb + (long long)xl * k1.sx, kL);
  ld12(kb + (long long)x * k1.sx, kC);
  ld12(kb + (long long)xr * k1.sx, kR);
  // transposed horizontal pass: position x receives from x' with clamp(x'+t-1) == x
  float g_v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a = gh[1][e] * kC[e][1];
    if (x + 1 < W) a += gh[2][e] * kR[e][0];
    if (x >= 1) a += gh[0][e] * kL[e][2];
    if (x == 0) a += gh[1][e] * kC[e][0];
    if (x == W - 1) a += gh[1][e] * kC[e][2];
    g_v[e] = a;
  }
  // kernel gradient: horizontal pass g_h * v[clamp(x+t-1)], vertical pass g_v * s[clamp(y+t-1)]
  float gK[4][3];
  const int ys[3] = {y > 0 ? y - 1 : 0, y, y < H - 1 ? y + 1 : H - 1};
#pragma unroll
  for (int tt = 0; tt < 3; ++tt) {
    const float4 vv = *reinterpret_cast<const float4*>(v + (rowb + xs[tt]) * C + cq * 4);
    const float4 ss = *reinterpret_cast<const float4*>(s + (((long long)b * H + ys[tt]) * W + x) * C + cq * 4);
    gK[0][tt] = gh[1][0] * vv.x + g_v[0] * ss.x; gK[1][tt] = gh[1][1] * vv.y + g_v[1] * ss.y;
    gK[2][tt] = gh[1][2] * vv.z + g_v[2] * ss.z; gK[3][tt] = gh[1][3] * vv.w + g_v[3] * ss.w;
  }
  float* gkp = gk.p + (long long)b * gk.sb + (long long)y * gk.sy + (long long)x * gk.sx + cq * 12;
  float4 o0 = make_float4(gK[0][0], gK[0][1], gK[0][2], gK[1][0]), o1 = make_float4(gK[1][1], gK[1][2], gK[2][0], gK[2][1]),
         o2 = make_float4(gK[2][2], gK[3][0], gK[3][1], gK[3][2]);
  if (k_accumulate) {
    const float4 p0 = *reinterpret_cast<const float4*>(gkp), p1 = *reinterpret_cast<const float4*>(gkp + 4), p2 = *reinterpret_cast<const float4*>(gkp + 8);
    o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w; o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
    o2.x += p2.x; o2.y += p2.y; o2.z += p2.z; o2.w += p2.w;
  }
  *reinterpret_cast<float4*>(gkp) = o0; *reinterpret_cast<float4*>(gkp + 4) = o1; *reinterpret_cast<float4*>(gkp + 8) = o2;
  const long long o = (rowb + x) * C + cq * 4;
  *reinterpret_cast<float4*>(gv + o) = make_float4(g_v[0], g_v[1], g_v[2], g_v[3]);
  float4 gf = make_float4(gh[1][0], gh[1][1], gh[1][2], gh[1][3]);
  if (fin_accumulate) {
    const float4 p = *reinterpret_cast<const float4*>(gfin + o);
    gf.x += p.x; gf.y += p.y; gf.z += p.z; gf.w += p.w;
  }
  *reinterpret_cast<float4*>(gfin + o) = gf;
}

// thread = (pixel, lane q of 16): channels q + 16 e, e < NE (a wave-instruction of the scatter then adds 16 consecutive floats per
// pixel).  gv, prev, gprev: dense (B,H,W,C); off: view with 2 channels; goff: dense (B,H,W,2).
template <int NE>
__global__ __launch_bounds__(256) void iac_bwd_warp_kernel(const float* __restrict__ gv, View k1, const float* __restrict__ prev, View off, int B, int H,
                                                           int W, float* __restrict__ gprev, float* __restrict__ goff) {
  constexpr int C = 16 * NE;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long npix = (long long)B * H * W;
  long long pixg = t >> 4;
  const bool live = pixg < npix;
  if (!live) pixg = npix - 1;                       // (keeps the 16-lane shuffles below defined for the tail)
  const int q = (int)(t & 15);
  const int x = (int)(pixg % W);
  const int y = (int)((pixg / W) % H);
  const int b = (int)(pixg / ((long long)W * H));
  // g_s = transposed vertical pass of g_v
  const int yu = y > 0 ? y - 1 : 0, yd = y < H - 1 ? y + 1 : H - 1;
  float gs[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int c = q + 16 * e;
    const float* kb = k1.p + (long long)b * k1.sb + (long long)x * k1.sx + c * 3;
    const float* kU = kb + (long long)yu * k1.sy, *kC = kb + (long long)y * k1.sy, *kD = kb + (long long)yd * k1.sy;
    const float gC = gv[(((long long)b * H + y) * W + x) * C + c];
    const float gU = gv[(((long long)b * H + yu) * W + x) * C + c];
    const float gD = gv[(((long long)b * H + yd) * W + x) * C + c];
    float a = gC * kC[1];
    if (y + 1 < H) a += gD * kD[0];
    if (y >= 1) a += gU * kU[2];
    if (y == 0) a += gC * kC[0];
    if (y == H - 1) a += gC * kC[2];
    gs[e] = a;
  }
  // sampling position (same arithmetic as warp_kernel)
  const float* op = off.p + (long long)b * off.sb + (long long)y * off.sy + (long long)x * off.sx;
  const float fx = (float)x + op[0];
  const float fy = (float)y + op[off.sc];
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float wx1 = fx - x0f, wy1 = fy - y0f;
  const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const bool sane = (fx > -2.f) && (fx < (float)W + 1.f) && (fy > -2.f) && (fy < (float)H + 1.f);
  const int x0 = sane ? (int)x0f : -4, y0 = sane ? (int)y0f : -4;
  float dsx = 0.f, dsy = 0.f;
  float p[2][2][NE];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int xi = x0 + dx, yi = y0 + dy;
      const bool in = xi >= 0 && xi < W && yi >= 0 && yi < H;
      const long long o = (((long long)b * H + (in ? yi : 0)) * W + (in ? xi : 0)) * C;
      const float w = (dy ? wy1 : wy0) * (dx ? wx1 : wx0);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const float pv = in ? prev[o + q + 16 * e] : 0.f;
        p[dy][dx][e] = pv;
        if (in && live) atomicAdd(gprev + o + q + 16 * e, w * gs[e]);
      }
    }
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    dsx += gs[e] * (wy0 * (p[0][1][e] - p[0][0][e]) + wy1 * (p[1][1][e] - p[1][0][e]));
    dsy += gs[e] * (wx0 * (p[1][0][e] - p[0][0][e]) + wx1 * (p[1][1][e] - p[0][1][e]));
  }
  dsx += __shfl_xor(dsx, 1); dsx += __shfl_xor(dsx, 2); dsx += __shfl_xor(dsx, 4); dsx += __shfl_xor(dsx, 8);
  dsy += __shfl_xor(dsy, 1); dsy += __shfl_xor(dsy, 2); dsy += __shfl_xor(dsy, 4); dsy += __shfl_xor(dsy, 8);
  if (q == 0 && live) *reinterpret_cast<float2*>(goff + pixg * 2) = make_float2(dsx, dsy);
}

}  // namespace fcvsr

using namespace fcvsr;

static bool k_ok(const fcvsr_view* v, int C) {
  return v && v->ptr && v->dtype == FCVSR_F32 && v->sc == 1 && v->c == 3 * C && v->sx % 4 == 0 && v->sy % 4 == 0 && v->sb % 4 == 0 &&
         ((uintptr_t)v->ptr % 16) == 0;
}

extern "C" int fcvsr_iac_bwd_sac(const float* gy, const float* yout, const float* v, const float* s, const fcvsr_view* k1, float slope, int B,
                                 int H, int W, int C, float* gfin, int fin_accumulate, float* gv, const fcvsr_view* gk, int k_accumulate,
                                 void* stream) {
  FCVSR_CHECK_ARG(gy && yout && v && s && gfin && gv, "null pointer");
  FCVSR_CHECK_ARG(C % 4 == 0 && k_ok(k1, C) && k_ok(gk, C), "k1 / gk: f32 views with 3*C contiguous channels, 16-byte aligned");
  const long long total = (long long)B * H * W * (C / 4);
  hipLaunchKernelGGL(iac_bwd_sac_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gy, yout, v, s, to_view(*k1), slope, B, H, W, C,
                     gfin, fin_accumulate, gv, to_view(*gk), k_accumulate);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_iac_bwd_warp(const float* gv, const fcvsr_view* k1, const float* prev, const fcvsr_view* off, int B, int H, int W, int C,
                                  float* gprev_zeroed, float* goff, void* stream) {
  FCVSR_CHECK_ARG(gv && prev && gprev_zeroed && goff, "null pointer");
  FCVSR_CHECK_ARG((C == 32 || C == 64) && k_ok(k1, C), "C in {32, 64}; k1: f32 view with 3*C contiguous channels");
  FCVSR_CHECK_ARG(off && off->ptr && off->c >= 2 && off->dtype == FCVSR_F32, "off needs 2 f32 channels");
  const long long total = (long long)B * H * W * 16;
  if (C == 64)
    hipLaunchKernelGGL(iac_bwd_warp_kernel<4>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gv, to_view(*k1), prev, to_view(*off), B, H, W,
                       gprev_zeroed, goff);
  else
    hipLaunchKernelGGL(iac_bwd_warp_kernel<2>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gv, to_view(*k1), prev, to_view(*off), B, H, W,
                       gprev_zeroed, goff);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
