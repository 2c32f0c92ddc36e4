// Direct f32 convolution for gfx950: one output pixel per lane, CO_T output channels per lane.
//
// Mapping: a wave's 64 lanes are 64 consecutive output pixels and share one group of CO_T output channels, so every
// weight address is wave-uniform -> the compiler fetches weights with scalar loads (s_load_dwordx*) into SGPRs and each
// v_fma takes one VGPR (input) and one SGPR (weight): no LDS, no weight traffic through the vector memory path.
// Inputs are NHWC: a lane reads its pixel's channels as 16-byte vectors (its own contiguous segment).
// Exact f32 (fmaf chain); used for skinny layers (Cin=7/21, Cout=1/3/4), odd kernel sizes (1..11) and as the f32
// reference path for every layer.  Replaces nn.Conv2d + bias + activation + residual adds + torch.cat + PixelShuffle.
#include "common.h"

namespace fcvsr {

struct ConvArgs {
  int n_src;
  View src[3];
  int B, H, W, Ho, Wo, kh, kw, stride, pad, cout, cout_pad, cin_total;
  const float* w;
  const float* bias;
  int act;
  float slope;
  const float* slope_ptr;
  int n_res;
  View res[2];
  float rs[2];
  View dst;
  int ps;
  int dst_vec;
};

template <int CO_T, bool VEC>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
  const long long total = (long long)a.B * a.Ho * a.Wo;
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= total) return;
  const int ox = (int)(pix % a.Wo);
  const int oy = (int)((pix / a.Wo) % a.Ho);
  const int b = (int)(pix / ((long long)a.Wo * a.Ho));
  const int co0 = blockIdx.y * CO_T;

  float acc[CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; ++j) acc[j] = 0.f;

  int cbase = 0;
  for (int s = 0; s < a.n_src; ++s) {
    const View sv = a.src[s];
    const float* sp = sv.p + (long long)b * sv.sb;
    for (int ky = 0; ky < a.kh; ++ky) {
      const int iy = oy * a.stride - a.pad + ky;
      for (int kx = 0; kx < a.kw; ++kx) {
        const int ix = ox * a.stride - a.pad + kx;
        const bool ok = (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
        const float* ip = sp + (long long)iy * sv.sy + (long long)ix * sv.sx;
        const float* wp = a.w + ((long long)(ky * a.kw + kx) * a.cin_total + cbase) * a.cout_pad + co0;
        if (VEC) {
          for (int ci = 0; ci < sv.c; ci += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4*>(ip + ci);
            const float* w0 = wp + (long long)ci * a.cout_pad;
#pragma unroll
            for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v.x, w0[j], acc[j]);
#pragma unroll
            for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v.y, w0[a.cout_pad + j], acc[j]);
#pragma unroll
            for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v.z, w0[2 * a.cout_pad + j], acc[j]);
#pragma unroll
            for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v.w, w0[3 * a.cout_pad + j], acc[j]);
          }
        } else {
          for (int ci = 0; ci < sv.c; ++ci) {
            float v = 0.f;
            if (ok) v = ip[(long long)ci * sv.sc];
            const float* w0 = wp + (long long)ci * a.cout_pad;
#pragma unroll
            for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v, w0[j], acc[j]);
          }
        }
      }
    }
    cbase += sv.c;
  }

  float slope = a.slope;
  if (a.act == FCVSR_ACT_PRELU) slope = a.slope_ptr[0];
  float outv[CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; ++j) {
    const int co = co0 + j;
    float v = acc[j];
    if (co < a.cout) {
      if (a.bias) v += a.bias[co];
      if (a.act == FCVSR_ACT_RELU) v = fmaxf(v, 0.f);
      else if (a.act == FCVSR_ACT_LEAKY || a.act == FCVSR_ACT_PRELU) v = v >= 0.f ? v : v * slope;
      for (int r = 0; r < a.n_res; ++r) {
        const View rv = a.res[r];
        v += a.rs[r] * rv.p[(long long)b * rv.sb + (long long)oy * rv.sy + (long long)ox * rv.sx + (long long)co * rv.sc];
      }
    }
    outv[j] = v;
  }
  const View d = a.dst;
  if (a.ps) {
#pragma unroll
    for (int j = 0; j < CO_T; ++j) {
      const int co = co0 + j;
      if (co < a.cout) {
        const int c2 = co >> 2, i = (co >> 1) & 1, jj = co & 1;
        d.p[(long long)b * d.sb + (long long)(2 * oy + i) * d.sy + (long long)(2 * ox + jj) * d.sx + (long long)c2 * d.sc] = outv[j];
      }
    }
  } else if (a.dst_vec && (CO_T % 4 == 0)) {
    float* dp = d.p + (long long)b * d.sb + (long long)oy * d.sy + (long long)ox * d.sx + co0;
#pragma unroll
    for (int j = 0; j < CO_T; j += 4) {
      if (co0 + j < a.cout)
        *reinterpret_cast<float4*>(dp + j) = make_float4(outv[j], outv[j + 1], outv[j + 2], outv[j + 3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < CO_T; ++j) {
      const int co = co0 + j;
      if (co < a.cout)
        d.p[(long long)b * d.sb + (long long)oy * d.sy + (long long)ox * d.sx + (long long)co * d.sc] = outv[j];
    }
  }
}

template <int CO_T>
static void launch(const ConvArgs& a, bool vec, hipStream_t st) {
  const long long total = (long long)a.B * a.Ho * a.Wo;
  dim3 grid(cdiv(total, 256), cdiv(a.cout, CO_T));
  if (vec) hipLaunchKernelGGL((conv_direct_kernel<CO_T, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_direct_kernel<CO_T, false>), grid, dim3(256), 0, st, a);
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_conv2d(const fcvsr_conv_desc* d, void* stream) {
  FCVSR_CHECK_ARG(d != nullptr, "null descriptor");
  FCVSR_CHECK_ARG(d->n_src >= 1 && d->n_src <= 3, "n_src must be 1..3");
  FCVSR_CHECK_ARG(d->n_res >= 0 && d->n_res <= 2, "n_res must be 0..2");
  FCVSR_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->cout > 0, "empty problem");
  FCVSR_CHECK_ARG(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->pad >= 0, "bad geometry");
  FCVSR_CHECK_ARG(d->weight != nullptr && d->cout_pad >= d->cout && d->cout_pad % 16 == 0, "bad packed weight");
  FCVSR_CHECK_ARG(!(d->act == FCVSR_ACT_PRELU) || d->slope_ptr != nullptr, "PReLU needs slope_ptr");
  FCVSR_CHECK_ARG(!d->pixel_shuffle || d->cout % 4 == 0, "pixel_shuffle needs cout%4==0");
  ConvArgs a;
  a.n_src = d->n_src;
  bool vec = true;
  int cin = 0;
  for (int s = 0; s < d->n_src; ++s) {
    FCVSR_CHECK_ARG(d->src[s].dtype == FCVSR_F32 && d->src[s].ptr && d->src[s].c > 0, "src must be f32, non-null");
    a.src[s] = to_view(d->src[s]);
    vec = vec && vec4_ok(d->src[s]);
    cin += d->src[s].c;
  }
  a.cin_total = cin;
  a.B = d->B; a.H = d->H; a.W = d->W;
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
  a.Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
  a.Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
  FCVSR_CHECK_ARG(a.Ho > 0 && a.Wo > 0, "empty output");
  a.cout = d->cout; a.cout_pad = d->cout_pad;
  a.w = (const float*)d->weight; a.bias = d->bias;
  a.act = d->act; a.slope = d->slope; a.slope_ptr = d->slope_ptr;
  a.n_res = d->n_res;
  for (int r = 0; r < d->n_res; ++r) {
    FCVSR_CHECK_ARG(d->res[r].dtype == FCVSR_F32 && d->res[r].ptr, "res must be f32, non-null");
    a.res[r] = to_view(d->res[r]);
    a.rs[r] = d->res_scale[r];
  }
  FCVSR_CHECK_ARG(d->dst.dtype == FCVSR_F32 && d->dst.ptr, "dst must be f32, non-null");
  a.dst = to_view(d->dst);
  a.ps = d->pixel_shuffle;
  a.dst_vec = (!a.ps && vec4_ok(d->dst) && d->cout % 4 == 0) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (d->cout >= 16) launch<16>(a, vec, st);
  else if (d->cout > 4) launch<8>(a, vec, st);
  else if (d->cout > 1) launch<4>(a, vec, st);
  else launch<1>(a, vec, st);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
