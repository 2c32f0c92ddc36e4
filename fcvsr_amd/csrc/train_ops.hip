// Training-path helpers (reference: every nn.Conv2d of CVSR_train/arch/CVSR_freq.py under loss.backward(),
// train_LD_freqCVSR_S_22.py:244-251): the per-step weight re-packing and the activation / bias halves of a convolution's
// backward as single launches (they were chains of 4-8 small torch kernels per layer and step).
#include "common.h"
#include "mfma_util.h"

namespace fcvsr {

// (cout, cin, kh, kw) f32 -> 16-bit [kh*kw][rows_pad][cols_pad], zero padded: the operand layout of fcvsr_conv2d_mfma.
// transposed = 0: rows = cout, cols = cin, tap = ky*kw + kx (forward).
// transposed = 1: rows = cin, cols = cout, tap flipped (the input-gradient convolution: dL/dx = conv(dL/dy, W^T flipped)).
template <bool BF16>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ dst, int cout, int cin,
                                                          int kk, int rows_pad, int cols_pad, int transposed) {
  const long long total = (long long)kk * rows_pad * cols_pad;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int col = (int)(i % cols_pad);
    const long long t = i / cols_pad;
    const int row = (int)(t % rows_pad), tap = (int)(t / rows_pad);
    const int co = transposed ? col : row, ci = transposed ? row : col;
    float v = 0.f;
    if (co < cout && ci < cin) v = w[((long long)co * cin + ci) * kk + (transposed ? kk - 1 - tap : tap)];
    uint16_t o;
    if (BF16) { const __bf16 b = (__bf16)v; o = __builtin_bit_cast(uint16_t, b); }
    else { const _Float16 hh = (_Float16)v; o = __builtin_bit_cast(uint16_t, hh); }
    dst[i] = o;
  }
}

// Every planned weight of a training pass in ONE launch (the per-weight launches were 205 per step, 0.74 ms): tab[item] = {src, dst,
// cout, cin, kk, rows_pad, cols_pad, transposed, first block}; a block packs kPwmElems consecutive elements of its item.
constexpr int kPwmElems = 2048;
template <bool BF16>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const long long* __restrict__ tab, int n_items) {
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {                                        // last item whose first block is <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid * 9 + 8] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* it = tab + lo * 9;
  const float* w = reinterpret_cast<const float*>(it[0]);
  uint16_t* dst = reinterpret_cast<uint16_t*>(it[1]);
  const int cout = (int)it[2], cin = (int)it[3], kk = (int)it[4], rows_pad = (int)it[5], cols_pad = (int)it[6], transposed = (int)it[7];
  const long long total = (long long)kk * rows_pad * cols_pad;
  const long long e0 = ((long long)blockIdx.x - it[8]) * kPwmElems;
  for (long long i = e0 + threadIdx.x; i < e0 + kPwmElems && i < total; i += 256) {
    const int col = (int)(i % cols_pad);
    const long long t = i / cols_pad;
    const int row = (int)(t % rows_pad), tap = (int)(t / rows_pad);
    const int co = transposed ? col : row, ci = transposed ? row : col;
    float v = 0.f;
    if (co < cout && ci < cin) v = w[((long long)co * cin + ci) * kk + (transposed ? kk - 1 - tap : tap)];
    uint16_t o;
    if (BF16) { const __bf16 b = (__bf16)v; o = __builtin_bit_cast(uint16_t, b); }
    else { const _Float16 hh = (_Float16)v; o = __builtin_bit_cast(uint16_t, hh); }
    dst[i] = o;
  }
}

// g_pre = g * act'(y) with y the activation's OUTPUT (LeakyReLU / ReLU: the sign of y is the sign of the pre-activation for a
// positive slope; for slope 0 the y == 0 entries take the zero-side derivative like torch's threshold_backward).  4 floats per lane.
__global__ __launch_bounds__(256) void act_bwd_kernel(const float4* __restrict__ g, const float4* __restrict__ y, float4* __restrict__ out,
                                                      float slope, long long n4) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 a = g[i], b = y[i];
    out[i] = make_float4(b.x > 0.f ? a.x : a.x * slope, b.y > 0.f ? a.y : a.y * slope, b.z > 0.f ? a.z : a.z * slope,
                         b.w > 0.f ? a.w : a.w * slope);
  }
}

// Column sums of a dense (npix, C) f32 matrix (the bias gradient of a convolution), deterministic two-stage:
// stage 1: block b sums rows [b*kCsRows, (b+1)*kCsRows) into part[b][C] (a thread owns one column and every (256 / cols)-th row of
// the block: coalesced across the columns, independent loads unrolled); stage 2: out[c] = sum_b part[b][c], 4 threads per column.
constexpr int kCsRows = 128, kCsMaxBlk = 1024;
static inline int colsum_rows(long long npix) {                 // rows per block: at least kCsRows, at most kCsMaxBlk blocks
  const long long r = (npix + kCsMaxBlk - 1) / kCsMaxBlk;
  return (int)(r > kCsRows ? r : kCsRows);
}
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ x, long long npix, int C, int rpb, float* __restrict__ part) {
  const long long r0 = (long long)blockIdx.x * rpb;
  const long long r1 = r0 + rpb < npix ? r0 + rpb : npix;
  __shared__ float sm[256];
  for (int cb = 0; cb < C; cb += 256) {
    const int cols = C - cb < 256 ? C - cb : 256;               // columns handled in this sweep
    int cpt = 1;
    while (cpt < cols) cpt <<= 1;                               // power of two >= cols (<= 256)
    const int nrl = 256 / cpt, rl = threadIdx.x / cpt, c = threadIdx.x % cpt;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
      const float* px = x + cb + c;
      long long rr = r0 + rl;
      for (; rr + 3ll * nrl < r1; rr += 4ll * nrl) {
        s0 += px[rr * C]; s1 += px[(rr + nrl) * C]; s2 += px[(rr + 2ll * nrl) * C]; s3 += px[(rr + 3ll * nrl) * C];
      }
      for (; rr < r1; rr += nrl) s0 += px[rr * C];
    }
    sm[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < cols) {
      float t = 0.f;
      for (int q = 0; q < nrl; ++q) t += sm[q * cpt + threadIdx.x];
      part[(long long)blockIdx.x * C + cb + threadIdx.x] = t;
    }
    __syncthreads();
  }
}
static thread_local int g_colsum_accumulate = 0;                // fcvsr_colsum_set_accumulate: out += instead of out = (see wgrad.hip)
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ part, int nblk, int C, float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;       // 4 threads per column, fixed combination order
  __shared__ float sm[4][64];
  float s = 0.f;
  if (c < C) {
    int b = q;
    for (; b + 12 < nblk; b += 16) {                             // four independent loads in flight, added in block order
      const float p0 = part[(long long)b * C + c], p1 = part[(long long)(b + 4) * C + c], p2 = part[(long long)(b + 8) * C + c],
                  p3 = part[(long long)(b + 12) * C + c];
      s += p0; s += p1; s += p2; s += p3;
    }
    for (; b < nblk; b += 4) s += part[(long long)b * C + c];
  }
  sm[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && c < C) {
    const float t = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
    out[c] = accumulate ? out[c] + t : t;
  }
}

// PReLU with ONE shared slope (nn.PReLU(), reference CVSR_freq.py:2590 and ConvBlk :349): y = x > 0 ? x : a x, slope read from device
// memory (a parameter: no host sync, capturable).  Backward: gx = x > 0 ? g : a g; ga = sum over x <= 0 of g x, two-stage.
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float4* __restrict__ x, const float* __restrict__ slope, float4* __restrict__ y, long long n4) {
  const float a = slope[0];
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = x[i];
    y[i] = make_float4(v.x > 0.f ? v.x : a * v.x, v.y > 0.f ? v.y : a * v.y, v.z > 0.f ? v.z : a * v.z, v.w > 0.f ? v.w : a * v.w);
  }
}
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float4* __restrict__ g, const float4* __restrict__ x, const float* __restrict__ slope,
                                                        float4* __restrict__ gx, float* __restrict__ part, long long n4) {
  const float a = slope[0];
  float acc = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = x[i], gg = g[i];
    gx[i] = make_float4(v.x > 0.f ? gg.x : a * gg.x, v.y > 0.f ? gg.y : a * gg.y, v.z > 0.f ? gg.z : a * gg.z, v.w > 0.f ? gg.w : a * gg.w);
    acc += (v.x > 0.f ? 0.f : v.x * gg.x) + (v.y > 0.f ? 0.f : v.y * gg.y) + (v.z > 0.f ? 0.f : v.z * gg.z) + (v.w > 0.f ? 0.f : v.w * gg.w);
  }
  __shared__ float sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(256) void sum_small_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float sm[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sm[0];
}

// Weight gradient of a 3x3 "same" convolution with ONE output channel (conv_last0, 64 -> 1 at 4H x 4W, reference :2607):
//   dw[ci][ky][kx] = sum_p gy[p] x[p + (ky-1, kx-1)][ci]  =  sum_q x[q][ci] gy[q - (ky-1, kx-1)]
// thread = (pixel slot, 4 channels): per input pixel ONE 16-byte load of x and 9 (cached) scalars of gy feed 36 accumulators; x is
// read once.  part[blk][C][9]; the slabs are added in order by colsum_stage2 (bit-reproducible).
__global__ __launch_bounds__(256) void wgrad_cout1_kernel(const float* __restrict__ x, const float* __restrict__ gy, int B, int H, int W, int C, int rows_per_blk,
                                                          float* __restrict__ part) {
  const int CQ = C / 4, cq = threadIdx.x % CQ, slot = threadIdx.x / CQ, nslot = 256 / CQ;
  const long long r0 = (long long)blockIdx.x * rows_per_blk, nrows = (long long)B * H;
  const long long r1 = r0 + rows_per_blk < nrows ? r0 + rows_per_blk : nrows;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (slot < nslot) {
    for (long long rr = r0; rr < r1; ++rr) {
      const int y = (int)(rr % H);
      const long long b = rr / H;
      for (int xx = slot; xx < W; xx += nslot) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + y) * W + xx) * C + cq * 4);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int py = y - (ky - 1), px = xx - (kx - 1);            // the output pixel whose tap (ky, kx) reads this input pixel
            const bool in = py >= 0 && py < H && px >= 0 && px < W;
            const float gg = in ? gy[(b * H + py) * W + px] : 0.f;
            acc[ky * 3 + kx].x += gg * v.x; acc[ky * 3 + kx].y += gg * v.y; acc[ky * 3 + kx].z += gg * v.z; acc[ky * 3 + kx].w += gg * v.w;
          }
      }
    }
  }
  __shared__ float sm[9 * 1024];                                         // [slot][C][9] floats: C * nslot = C * 256 / (C / 4) = 1024
  float* mine = sm + ((long long)slot * 9) * C + cq * 36;               // [slot][ci][tap]: nn.Conv2d's (ci, ky, kx) order
  if (slot < nslot)
#pragma unroll
    for (int t = 0; t < 9; ++t) { mine[t] = acc[t].x; mine[9 + t] = acc[t].y; mine[18 + t] = acc[t].z; mine[27 + t] = acc[t].w; }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * C; i += 256) {
    float sacc = 0.f;
    for (int q = 0; q < nslot; ++q) sacc += sm[(long long)q * 9 * C + i];
    part[(long long)blockIdx.x * 9 * C + i] = sacc;
  }
}

// Backward of the CorrBlock lookup (fcvsr_corr_lookup, reference :1279-1337): corr[c = i*n + j][y][x] = x1f[s] * x2f[s] / sqrt(C) at the
// NHWC address s of flat element (y*Wf + x)*C + (y+j-r)*2 + (x+i-r) of the NCHW-contiguous product buffer.  For a fixed pixel the map
// (i, j) -> s is one-to-one and different pixels own different I_p, so every s is written at most once: plain stores into zeroed
// gradients, no atomics.  g: (B, H, xw-strip of Wf, gc) view; gx1 / gx2: dense NHWC (B, H*Wf, ps) zero-initialised.
__global__ __launch_bounds__(256) void corr_lookup_bwd_kernel(const float* __restrict__ x1f, const float* __restrict__ x2f, long long ps, int B, int H,
                                                              int Wf, int C, int radius, int xw, View g, float norm_div,
                                                              float* __restrict__ gx1, float* __restrict__ gx2) {
  const int n = 2 * radius + 1, nn = n * n;
  const long long total = (long long)B * H * xw * nn;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % nn);
  const long long pixg = t / nn;
  const int x = (int)(pixg % xw);
  const int y = (int)((pixg / xw) % H);
  const int b = (int)(pixg / ((long long)xw * H));
  const int i = c / n, j = c % n;
  const int col = x + i - radius, row = y + j - radius;
  if (col < 0 || col > 1 || row < 0 || row >= C / 2) return;
  const long long HW = (long long)H * Wf;
  const long long e = ((long long)y * Wf + x) * C + row * 2 + col;
  const int ch = (int)(e / HW);
  const long long pp = e % HW;
  const long long src = ((long long)b * HW + pp) * ps + ch;
  const float gg = g.p[(long long)b * g.sb + (long long)y * g.sy + (long long)x * g.sx + (long long)c * g.sc] / norm_div;
  gx1[src] = gg * x2f[src];
  gx2[src] = gg * x1f[src];
}

}  // namespace fcvsr

using namespace fcvsr;

extern "C" int fcvsr_pack_weight_mfma(const float* w, int cout, int cin, int kh, int kw, void* dst, int rows_pad, int cols_pad,
                                      int dtype, int transposed, void* stream) {
  FCVSR_CHECK_ARG(w && dst, "null pointer");
  FCVSR_CHECK_ARG(dtype == FCVSR_BF16 || dtype == FCVSR_F16, "16-bit destination only");
  FCVSR_CHECK_ARG(rows_pad >= (transposed ? cin : cout) && cols_pad >= (transposed ? cout : cin), "padding smaller than the tensor");
  const long long total = (long long)kh * kw * rows_pad * cols_pad;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  if (dtype == FCVSR_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)dst, cout, cin, kh * kw, rows_pad, cols_pad, transposed);
  else
    hipLaunchKernelGGL(pack_weight_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)dst, cout, cin, kh * kw, rows_pad, cols_pad, transposed);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

/* fcvsr_pack_weight_mfma for n_items weights in one launch.  tab (device memory, 9 x int64 per item): source pointer (f32, contiguous
 * (cout,cin,kh,kw)), destination pointer, cout, cin, kh*kw, rows_pad, cols_pad, transposed, first block; item i owns the blocks
 * [first_i, first_i + ceil(kh*kw*rows_pad*cols_pad / fcvsr_pack_weights_multi_block_elems())); total_blocks = their sum. */
extern "C" int fcvsr_pack_weights_multi_block_elems(void) { return kPwmElems; }
extern "C" int fcvsr_pack_weights_mfma_multi(const long long* tab, int n_items, int total_blocks, int dtype, void* stream) {
  FCVSR_CHECK_ARG(tab && n_items >= 1 && total_blocks >= 1, "empty plan");
  FCVSR_CHECK_ARG(dtype == FCVSR_BF16 || dtype == FCVSR_F16, "16-bit destination only");
  if (dtype == FCVSR_BF16) hipLaunchKernelGGL(pack_weight_multi_kernel<true>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, tab, n_items);
  else hipLaunchKernelGGL(pack_weight_multi_kernel<false>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, tab, n_items);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_act_bwd(const float* g, const float* y, float* out, float slope, long long n, void* stream) {
  FCVSR_CHECK_ARG(g && y && out, "null pointer");
  FCVSR_CHECK_ARG(n % 4 == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)out % 16) == 0, "16-byte aligned, n % 4 == 0");
  const long long n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  if (n4 > 0) hipLaunchKernelGGL(act_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)g, (const float4*)y, (float4*)out, slope, n4);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" long long fcvsr_colsum_scratch_elems(long long npix, int C) {
  const int rpb = colsum_rows(npix);
  const long long nblk = (npix + rpb - 1) / rpb;
  return (nblk > 0 ? nblk : 1) * C;
}

extern "C" int fcvsr_colsum(const float* x, long long npix, int C, float* out, float* scratch, long long scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(x && out && scratch, "null pointer");
  FCVSR_CHECK_ARG(C >= 1 && npix >= 1, "empty matrix");
  const int rpb = colsum_rows(npix);
  const long long nblk = (npix + rpb - 1) / rpb;
  FCVSR_CHECK_ARG(scratch_elems >= nblk * C, "scratch too small");
  hipLaunchKernelGGL(colsum_stage1, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, npix, C, rpb, scratch);
  hipLaunchKernelGGL(colsum_stage2, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, scratch, (int)nblk, C, out, g_colsum_accumulate);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_prelu_fwd(const float* x, const float* slope, float* y, long long n, void* stream) {
  FCVSR_CHECK_ARG(x && slope && y, "null pointer");
  FCVSR_CHECK_ARG(n % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "16-byte aligned, n % 4 == 0");
  const long long n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  if (n4 > 0) hipLaunchKernelGGL(prelu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)x, slope, (float4*)y, n4);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

/* gx = dL/dx, gslope[0] = dL/dslope; scratch >= 2048 floats */
extern "C" int fcvsr_prelu_bwd(const float* g, const float* x, const float* slope, float* gx, float* gslope, float* scratch, long long n,
                               void* stream) {
  FCVSR_CHECK_ARG(g && x && slope && gx && gslope && scratch, "null pointer");
  FCVSR_CHECK_ARG(n % 4 == 0 && n > 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)gx % 16) == 0, "16-byte aligned, n % 4 == 0");
  const long long n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)g, (const float4*)x, slope, (float4*)gx, scratch, n4);
  hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, grid, gslope);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" long long fcvsr_wgrad_cout1_scratch_elems(int B, int H, int C) {
  const long long nrows = (long long)B * H;
  const long long nblk = nrows < 1024 ? nrows : 1024;
  return nblk * 9 * C;
}

/* dw (1, C, 3, 3) of a 3x3 "same" convolution with one output channel; x dense (B,H,W,C) f32, gy dense (B,H,W) f32; C in {16, 32, 64} */
extern "C" int fcvsr_wgrad_cout1(const float* x, const float* gy, int B, int H, int W, int C, float* dw, float* scratch, long long scratch_elems,
                                 void* stream) {
  FCVSR_CHECK_ARG(x && gy && dw && scratch, "null pointer");
  FCVSR_CHECK_ARG(C == 16 || C == 32 || C == 64, "C in {16, 32, 64}");
  const long long nrows = (long long)B * H;
  const int nblk = (int)(nrows < 1024 ? nrows : 1024);
  const int rpb = (int)((nrows + nblk - 1) / nblk);
  const int nb = (int)((nrows + rpb - 1) / rpb);
  FCVSR_CHECK_ARG(scratch_elems >= (long long)nb * 9 * C, "scratch too small");
  FCVSR_CHECK_ARG(((uintptr_t)x % 16) == 0, "x 16-byte aligned");
  hipLaunchKernelGGL(wgrad_cout1_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, gy, B, H, W, C, rpb, scratch);
  // column sums of the (nb, C*9) partial matrix in block order = dw in nn.Conv2d's (1, C, 3, 3) layout
  hipLaunchKernelGGL(colsum_stage2, dim3((9 * C + 63) / 64), dim3(256), 0, (hipStream_t)stream, scratch, nb, 9 * C, dw, g_colsum_accumulate);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

/* column sums over 1..3 dense (npix[g], C) matrices added together (the bias gradient of a layer applied to several pyramid levels):
 * stage 1 per matrix into consecutive block ranges of one scratch buffer, ONE ordered stage 2 */
extern "C" long long fcvsr_colsum_groups_scratch_elems(const long long* npix, int n_groups, int C) {
  long long nblk = 0;
  for (int g = 0; g < n_groups; ++g) { const int rpb = colsum_rows(npix[g]); nblk += (npix[g] + rpb - 1) / rpb; }
  return (nblk > 0 ? nblk : 1) * C;
}

extern "C" int fcvsr_colsum_groups(const float* const* xs, const long long* npix, int n_groups, int C, float* out, float* scratch,
                                   long long scratch_elems, void* stream) {
  FCVSR_CHECK_ARG(xs && npix && out && scratch && n_groups >= 1 && n_groups <= 3 && C >= 1, "bad arguments");
  long long blk0 = 0;
  for (int g = 0; g < n_groups; ++g) {
    FCVSR_CHECK_ARG(xs[g] && npix[g] >= 1, "empty matrix");
    const int rpb = colsum_rows(npix[g]);
    const long long nblk = (npix[g] + rpb - 1) / rpb;
    FCVSR_CHECK_ARG(scratch_elems >= (blk0 + nblk) * C, "scratch too small");
    hipLaunchKernelGGL(colsum_stage1, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, xs[g], npix[g], C, rpb, scratch + blk0 * C);
    blk0 += nblk;
  }
  hipLaunchKernelGGL(colsum_stage2, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, scratch, (int)blk0, C, out, g_colsum_accumulate);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" void fcvsr_colsum_set_accumulate(int on) { g_colsum_accumulate = on ? 1 : 0; }

/* backward of fcvsr_corr_lookup: g = dL/dcorr restricted to the first x_count columns (view with >= (2r+1)^2 channels), x1f / x2f as in
 * the forward; gx1 / gx2: dense (B, H, Wf, pix_stride) f32 tensors that the caller has ZEROED (every element is written at most once) */
extern "C" int fcvsr_corr_lookup_bwd(const float* x1f, const float* x2f, int64_t pix_stride, int B, int H, int Wf, int C, int radius, int x_count,
                                     const fcvsr_view* g, float* gx1_zeroed, float* gx2_zeroed, void* stream) {
  FCVSR_CHECK_ARG(x1f && x2f && g && g->ptr && gx1_zeroed && gx2_zeroed, "null pointer");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && Wf > 0 && C > 0 && C % 2 == 0 && radius >= 0 && pix_stride >= C, "bad sizes");
  FCVSR_CHECK_ARG(g->c >= (2 * radius + 1) * (2 * radius + 1) && g->dtype == FCVSR_F32, "g: f32, >= (2r+1)^2 channels");
  FCVSR_CHECK_ARG(x_count > 0 && x_count <= Wf, "x_count must be in [1, Wf]");
  const long long total = (long long)B * H * x_count * (2 * radius + 1) * (2 * radius + 1);
  hipLaunchKernelGGL(corr_lookup_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x1f, x2f, (long long)pix_stride, B, H, Wf, C,
                     radius, x_count, to_view(*g), sqrtf((float)C), gx1_zeroed, gx2_zeroed);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
