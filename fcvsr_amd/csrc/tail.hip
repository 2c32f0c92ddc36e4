// Up-sampler tail helpers (reference CVSR_freq.py:2633-2645): PixelShuffle(2) of a dense NHWC tensor and the x4 bilinear
// base skip F.interpolate(shortcut[:, T//2], scale_factor=4, mode='bilinear') (align_corners=False).
#include "common.h"

namespace fcvsr {

// dst[b][2h+i][2w+j][c] = src[b][h][w][4c+2i+j]
__global__ void pixel_shuffle_kernel(const float* src, float* dst, int B, int H, int W, int C) {
  const long long total = (long long)B * H * W * C;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int ci = (int)(t % C);
  const long long pg = t / C;
  const int x = (int)(pg % W);
  const int y = (int)((pg / W) % H);
  const int b = (int)(pg / ((long long)W * H));
  const int c2 = ci >> 2, i = (ci >> 1) & 1, j = ci & 1;
  const int Co = C / 4;
  dst[(((long long)b * 2 * H + 2 * y + i) * 2 * W + 2 * x + j) * Co + c2] = src[t];
}

__global__ void bilinear_up4_kernel(View src, int B, int H, int W, View dst) {
  const int Ho = 4 * H, Wo = 4 * W;
  const long long total = (long long)B * Ho * Wo * dst.c;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  // x fastest so that NCHW destinations are written coalesced
  const int ox = (int)(t % Wo);
  const int oy = (int)((t / Wo) % Ho);
  const int c = (int)((t / ((long long)Wo * Ho)) % dst.c);
  const int b = (int)(t / ((long long)Wo * Ho * dst.c));
  float sy = 0.25f * ((float)oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
  float sx = 0.25f * ((float)ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float* sp = src.p + (long long)b * src.sb + (long long)c * src.sc;
  const float v00 = sp[(long long)y0 * src.sy + (long long)x0 * src.sx], v01 = sp[(long long)y0 * src.sy + (long long)x1 * src.sx];
  const float v10 = sp[(long long)y1 * src.sy + (long long)x0 * src.sx], v11 = sp[(long long)y1 * src.sy + (long long)x1 * src.sx];
  const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  dst.p[(long long)b * dst.sb + (long long)oy * dst.sy + (long long)ox * dst.sx + (long long)c * dst.sc] = v;
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_pixel_shuffle(const float* src, float* dst, int B, int H, int W, int C, void* stream) {
  FCVSR_CHECK_ARG(src && dst, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "C%4==0 required");
  const long long total = (long long)B * H * W * C;
  hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, B, H, W, C);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_bilinear_up4(const fcvsr_view* src, int B, int H, int W, const fcvsr_view* dst, void* stream) {
  FCVSR_CHECK_ARG(src && dst && src->ptr && dst->ptr, "null pointer");
  FCVSR_CHECK_ARG(src->dtype == FCVSR_F32 && dst->dtype == FCVSR_F32, "f32 only");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0 && src->c == dst->c && dst->c > 0, "bad sizes");
  const long long total = (long long)B * 16 * H * W * dst->c;
  hipLaunchKernelGGL(bilinear_up4_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, to_view(*src), B, H, W,
                     to_view(*dst));
  FCVSR_LAUNCH_CHECK();
  return 0;
}
