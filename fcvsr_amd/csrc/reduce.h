// Deterministic two-stage per-(batch, channel) reductions over pixels (no float atomics: fixed summation order).
#pragma once
#include "common.h"

namespace fcvsr {

constexpr int kRedThreads = 256;
constexpr int kRedPix = 256;   // pixels per stage-1 block (enough blocks to fill 256 CUs at 180x320)

inline int red_blocks(long long npix) { return cdiv(npix, kRedPix); }

// Stage 1: partial[k][b][blk][c] = sum over this block's pixels of f(b, pix, c)[k].
// Thread (sub, c): sub = tid / C strides over pixels; lanes with consecutive tid read consecutive channels.
template <int K, class F>
__global__ __launch_bounds__(kRedThreads) void reduce_stage1(F f, int B, long long npix, int C, float* partial) {
  __shared__ float sm[K][kRedThreads];
  const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int R = kRedThreads / C;  // pixel sub-lanes (C <= 256)
  const int sub = threadIdx.x / C, c = threadIdx.x % C;
  float acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.f;
  if (sub < R) {
    const long long p0 = (long long)blk * kRedPix;
    const long long p1 = (p0 + kRedPix < npix) ? p0 + kRedPix : npix;
    // eight pixels per step: their loads are issued together, the additions keep the ascending-pixel order
    long long p = p0 + sub;
    for (; K == 1 && p + 7 * R < p1; p += 8 * R) {        // (the two-value DivEnh expression measured slower unrolled: K == 1 only)
      float v[8][K];
#pragma unroll
      for (int u = 0; u < 8; ++u) f(b, p + (long long)u * R, c, v[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += v[u][k];
    }
    for (; p < p1; p += R) {
      float v[K];
      f(b, p, c, v);
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) sm[k][threadIdx.x] = acc[k];
  __syncthreads();
  if (threadIdx.x < C) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float s = 0.f;
      for (int r = 0; r < R; ++r) s += sm[k][r * C + threadIdx.x];
      partial[(((long long)k * B + b) * nblk + blk) * C + threadIdx.x] = s;
    }
  }
}

// Stage 2: out[kb][c] = sum_blk partial[kb][blk][c]   (kb in [0, K*B)); one 256-thread block per kb, fixed order
static __global__ __launch_bounds__(kRedThreads) void reduce_stage2(const float* partial, int KB, int nblk, int C,
                                                                    float* out) {
  __shared__ float sm[kRedThreads];
  const int kb = blockIdx.x;
  const int R = kRedThreads / C;
  const int sub = threadIdx.x / C, c = threadIdx.x % C;
  float s = 0.f;
  if (sub < R) {
    // eight loads in flight per step, added in ascending block order (the rolled loop waited for every single load)
    const float* pp = partial + (long long)kb * nblk * C + c;
    int blk = sub;
    for (; blk + 7 * R < nblk; blk += 8 * R) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pp[(long long)(blk + u * R) * C];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; blk < nblk; blk += R) s += pp[(long long)blk * C];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int q = 0; q < R; ++q) t += sm[q * C + threadIdx.x];
    out[(long long)kb * C + threadIdx.x] = t;
  }
}

}  // namespace fcvsr
