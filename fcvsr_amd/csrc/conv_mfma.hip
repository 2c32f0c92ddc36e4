// Implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_{bf16,f16}), im2col-free.
//
//   GEMM view:  D[pixel][cout] = sum_{tap, cin} A[pixel + tap][cin] * W[tap][cout][cin]      (f32 accumulate)
//
// Workgroup = 256 threads (4 waves) -> output tile of 8 rows x 32 pixels x NT output channels.
//   * The (8+2p) x (32+2p) input halo tile of one 64-channel chunk is staged ONCE into LDS (converted f32 -> 16-bit while
//     staging; activations stay f32 in HBM so the residual trunk never accumulates 16-bit rounding) and is re-read for all
//     k*k taps: the 9 shifted A operands of a 3x3 conv are just 9 different LDS base addresses (no im2col buffer).
//   * A wave owns 2 tile rows (= 2 MFMA M-fragments of 32 pixels) x NT couts (NT/32 N-fragments): (2 + NT/32) ds_read_b128
//     feed 2*NT/32 MFMAs per 16-deep k-step.
//   * LDS rows are padded by 16 bytes (72 halfwords per 64-channel pixel) which makes every ds_read_b128 of a fragment
//     (32 consecutive pixels / couts, same k) conflict-free: bank step = 36 dwords -> 16 distinct 4-bank slots per lane group.
//   * Weights ([tap][cout][cin], 16-bit, cin contiguous) stream tap by tap: global -> registers (prefetched under the MFMAs
//     of the previous tap) -> LDS.  LDS footprint 67 KiB (NT<=128) -> 2 workgroups per CU overlap each other's stalls.
//   * Epilogue straight from the accumulators: bias, ReLU/LeakyReLU/PReLU, up to two scaled residual inputs, optional
//     PixelShuffle(2) scatter; a half-wave stores 32 consecutive couts of one pixel = one 128-byte line.
//   * 1x1 convolutions run in "flat" mode: the B*H*W pixels are tiled as a 1-D list, so odd widths (Wf = W/2+1) cost nothing.
//   * Up to 3 "groups" (e.g. the three pyramid levels of BlockRCB, which share weights) run in ONE launch so that the small
//     levels do not leave most of the 256 CUs idle.
// Replaces nn.Conv2d(+bias+activation+residual+cat+PixelShuffle) for every stride-1 layer of the path (see fcvsr_hip.h).
#include <stdlib.h>
#include "common.h"
#include "mfma_util.h"
#include "conv_res.h"

namespace fcvsr {

constexpr int kTW = 32, kCK = 64, kLD = kCK + 8;  // tile cols, channel chunk, padded LDS row (halfwords); tile rows = 4*MW

struct MGroup {
  View src[3];
  View res[2];
  View dst;
  float* gc_partial;   // [B][tiles_per_image*4][cout+2] (nullptr = off)
  int B, H, W;         // spatial size (stride-1 "same" conv: output size == input size)
  int tiles_x, tiles_y;
  int tile_begin;      // first flattened tile id of this group
};

struct MfmaArgs {
  int n_groups;
  MGroup g[3];
  int n_src, n_res;
  int seg_c[3];        // channels per source segment
  int cin_total, cin16, cin_pad, cout, cout_pad, n_nblk;
  const uint16_t* w;   // [taps][cout_pad][cin_pad]
  const float* bias;
  int act;
  float slope;
  const float* slope_ptr;
  float rs[2];
  int ps;
  int flat;            // 1x1: treat pixels as a flat list of B*H*W
  int src16, dst16;    // sources / destination stored in the MFMA dtype (16-bit) instead of f32
  int dstbf;           // the 16-bit destination format is bf16 (generic kernel: may differ from the MFMA dtype)
  int res16;           // residual inputs stored in the MFMA dtype (lean 3x3 kernel only: 16-bit trunk)
  int gc16;            // lean 3x3 kernel with ContextBlock fusion: the 4-couts-per-lane epilogue stores the MFMA dtype (8 bytes)
  const float* gc_wmask;   // ContextBlock fusion: per-wave online-softmax partials of the output (cout <= 64, 3x3)
  int planar;          // single f32 source with arbitrary channel stride (the NCHW frames of feat_extract), cin <= 64
  int sub2;            // stride-2 convolution: evaluate at full resolution, keep the even output pixels only
  int dbg;             // ablation switches for profiling builds (FCVSR_MFMA_DBG): 1 skip staging, 2 skip MFMA, 4 skip stores, 8 skip weight loads
};

struct EpiCtx {
  int act, n_res, ps, flat, H, W, b, dst16, dstbf, sub2, cq4;   // dstbf: 16-bit destination is bf16 (may differ from the MFMA dtype)   // cq4 = cout/4 (pixel-shuffle: couts are ordered sub-pixel-major)
  float slope, rs0, rs1;
  long long npix;
  View res0, res1, dst;
};

// Epilogue for one pixel x 4 consecutive output channels (n..n+3) held as a float4 (after the LDS transpose).
// Element offsets are 32-bit (host checks every tensor spans < 2^31 elements).
template <bool BF16>
__device__ __forceinline__ bool epilogue_quad(const EpiCtx& e, float4 v, const float* bias, int cout, int n, int py, int px,
                                              long long pflat, float4* xo) {
  bool pok = e.flat ? (pflat < e.npix) : ((py < e.H) && (px < e.W));
  if (e.sub2) {                      // stride 2: only even positions exist in the output; res/dst are at half resolution
    pok = pok && !((py | px) & 1);
    py >>= 1;
    px >>= 1;
  }
  if (!pok || n >= cout) return false;
  const bool full = (n + 3 < cout);
  float x[4] = {v.x, v.y, v.z, v.w};
  if (bias) {
    if (full) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
      x[0] += b4.x; x[1] += b4.y; x[2] += b4.z; x[3] += b4.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (n + q < cout) x[q] += bias[n + q];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    x[q] = x[q] >= 0.f ? x[q] : x[q] * e.slope;        // e.slope = negative-side factor: 0 (ReLU), the slope, or 1 (none) - no branch on act
  }
  // residual inputs are indexed by the ORIGINAL output channel: with pixel-shuffle packing row n = sp*(cout/4)+c is
  // original channel 4c+sp
  const int sp_r = e.ps ? n / e.cq4 : 0;
  const int c_r = e.ps ? n - sp_r * e.cq4 : n;
#pragma unroll
  for (int ri = 0; ri < 2; ++ri) {
    if (ri < e.n_res) {
      const View& rv = ri == 0 ? e.res0 : e.res1;
      const float rs = ri == 0 ? e.rs0 : e.rs1;
      const int o = (e.flat ? (int)pflat * (int)rv.sx : (e.b * (int)rv.sb + py * (int)rv.sy + px * (int)rv.sx));
      if (full && rv.sc == 1 && !e.ps) {
        const float4 r4 = *reinterpret_cast<const float4*>(rv.p + o + n);
        x[0] = fmaf(rs, r4.x, x[0]); x[1] = fmaf(rs, r4.y, x[1]); x[2] = fmaf(rs, r4.z, x[2]); x[3] = fmaf(rs, r4.w, x[3]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = e.ps ? 4 * (c_r + q) + sp_r : n + q;
          if (n + q < cout) x[q] = fmaf(rs, rv.p[o + ch * (int)rv.sc], x[q]);
        }
      }
    }
  }
  const View& d = e.dst;
  int o;
  int nn = n;
  if (e.ps) {
    // pixel-shuffle: the host orders output channels sub-pixel-major, n = (2*i+j)*(cout/4) + c, so a lane's 4 consecutive
    // couts are 4 consecutive channels c of the same sub-pixel (i,j): one vector store at pixel (2y+i, 2x+j).
    int qy = py, qx = px, qb = e.b;
    if (e.flat) {
      qx = (int)(pflat % e.W);
      qy = (int)((pflat / e.W) % e.H);
      qb = (int)(pflat / ((long long)e.W * e.H));
    }
    const int sp = n / e.cq4;
    nn = n - sp * e.cq4;
    o = qb * (int)d.sb + (2 * qy + (sp >> 1)) * (int)d.sy + (2 * qx + (sp & 1)) * (int)d.sx;
  } else {
    o = (e.flat ? (int)pflat * (int)d.sx : (e.b * (int)d.sb + py * (int)d.sy + px * (int)d.sx));
  }
  if (e.dst16) {
    uint16_t* dp = reinterpret_cast<uint16_t*>(d.p);
    const uint2 pk = e.dstbf ? cvt4<true>(make_float4(x[0], x[1], x[2], x[3])) : cvt4<false>(make_float4(x[0], x[1], x[2], x[3]));
    if (full && d.sc == 1) {
      *reinterpret_cast<uint2*>(dp + o + nn) = pk;
    } else {
      const uint16_t hv[4] = {(uint16_t)(pk.x & 0xffff), (uint16_t)(pk.x >> 16), (uint16_t)(pk.y & 0xffff), (uint16_t)(pk.y >> 16)};
#pragma unroll
      for (int q = 0; q < 4; ++q) if (n + q < cout) dp[o + (nn + q) * (int)d.sc] = hv[q];
    }
  } else if (full && d.sc == 1) {
    *reinterpret_cast<float4*>(d.p + o + nn) = make_float4(x[0], x[1], x[2], x[3]);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) if (n + q < cout) d.p[o + (nn + q) * (int)d.sc] = x[q];
  }
  *xo = make_float4(x[0], n + 1 < cout ? x[1] : 0.f, n + 2 < cout ? x[2] : 0.f, n + 3 < cout ? x[3] : 0.f);
  return true;
}

// Fast path for 16-bit destinations: one pixel x 8 consecutive output channels per lane = one 16-byte store
// (no pixel shuffle, channel-contiguous dst, cout % 8 == 0; residuals are f32 views).
template <bool BF16>
__device__ __forceinline__ void epilogue_oct(const EpiCtx& e, float4 va, float4 vb, const float* bias, int n, int py, int px,
                                             long long pflat) {
  bool pok = e.flat ? (pflat < e.npix) : ((py < e.H) && (px < e.W));
  if (e.sub2) {
    pok = pok && !((py | px) & 1);
    py >>= 1;
    px >>= 1;
  }
  if (!pok) return;
  float x[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
  if (bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + n), b1 = *reinterpret_cast<const float4*>(bias + n + 4);
    x[0] += b0.x; x[1] += b0.y; x[2] += b0.z; x[3] += b0.w; x[4] += b1.x; x[5] += b1.y; x[6] += b1.z; x[7] += b1.w;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    x[q] = x[q] >= 0.f ? x[q] : x[q] * e.slope;        // e.slope = negative-side factor: 0 (ReLU), the slope, or 1 (none) - no branch on act
  }
#pragma unroll
  for (int ri = 0; ri < 2; ++ri) {
    if (ri < e.n_res) {
      const View& rv = ri == 0 ? e.res0 : e.res1;
      const float rs = ri == 0 ? e.rs0 : e.rs1;
      const int o = (e.flat ? (int)pflat * (int)rv.sx : (e.b * (int)rv.sb + py * (int)rv.sy + px * (int)rv.sx));
      if (rv.sc == 1) {
        const float4 r0 = *reinterpret_cast<const float4*>(rv.p + o + n), r1 = *reinterpret_cast<const float4*>(rv.p + o + n + 4);
        x[0] = fmaf(rs, r0.x, x[0]); x[1] = fmaf(rs, r0.y, x[1]); x[2] = fmaf(rs, r0.z, x[2]); x[3] = fmaf(rs, r0.w, x[3]);
        x[4] = fmaf(rs, r1.x, x[4]); x[5] = fmaf(rs, r1.y, x[5]); x[6] = fmaf(rs, r1.z, x[6]); x[7] = fmaf(rs, r1.w, x[7]);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = fmaf(rs, rv.p[o + (n + q) * (int)rv.sc], x[q]);
      }
    }
  }
  const View& d = e.dst;
  const int o = (e.flat ? (int)pflat * (int)d.sx : (e.b * (int)d.sb + py * (int)d.sy + px * (int)d.sx));
  uint2 lo, hi;
  if (e.dstbf) { lo = cvt4<true>(make_float4(x[0], x[1], x[2], x[3])); hi = cvt4<true>(make_float4(x[4], x[5], x[6], x[7])); }
  else { lo = cvt4<false>(make_float4(x[0], x[1], x[2], x[3])); hi = cvt4<false>(make_float4(x[4], x[5], x[6], x[7])); }
  *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(d.p) + o + n) = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// a wave-uniform 64-bit value pinned to scalar registers (so that a per-lane select between such values stays a v_cndmask
// and is not folded back into a load through a selected address)
__device__ __forceinline__ long long uniform64(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long sel3(int i, long long a, long long b, long long c) { return i == 0 ? a : (i == 1 ? b : c); }
// loads through a pointer rebuilt from integers must be typed global: generic pointers compile to flat_load, which also
// turns every counted lgkmcnt wait of the kernel into a full one
typedef unsigned __attribute__((ext_vector_type(4))) cm_u32x4_t;
typedef float __attribute__((ext_vector_type(4))) cm_f32x4_t;
typedef __attribute__((address_space(1))) cm_u32x4_t cm_guint4_t;
typedef __attribute__((address_space(1))) cm_f32x4_t cm_gfloat4_t;
__device__ __forceinline__ uint4 gld_u4(long long addr) { const cm_u32x4_t v = *reinterpret_cast<const cm_guint4_t*>((unsigned long long)addr); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 gld_f4(long long addr) { const cm_f32x4_t v = *reinterpret_cast<const cm_gfloat4_t*>((unsigned long long)addr); return make_float4(v.x, v.y, v.z, v.w); }

template <bool BF16, int NT, int KS, int MW, bool WD>
__global__ __launch_bounds__(256, MW == 1 ? (WD ? (NT == 128 ? 2 : 3) : (NT == 128 ? 3 : 4)) : 2) void conv_mfma_kernel(MfmaArgs a) {
  constexpr bool FLAT = KS == 1;                     // the host sets a.flat exactly for the 1x1 kernels
  constexpr int kTH = 4 * MW;                       // MW tile rows (M-fragments) per wave
  constexpr int PAD = KS / 2;
  constexpr int HH = kTH + 2 * PAD, HWD = kTW + 2 * PAD;
  constexpr int NF = NT / 32;
  constexpr int WLOADS = NT * kCK * 2 / 16 / 256;   // 16-byte weight loads per thread per tap (NT/32)
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* A_s = lds;                               // [HH*HWD][kLD]
  uint16_t* B_s = lds + HH * HWD * kLD;              // [NT][kLD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // ---- which tile ----------------------------------------------------------------------------------------------------
  // XCD-aware remap (workgroups are dealt round-robin over the 8 XCDs, each with a private L2): give every XCD one
  // contiguous run of the work list so that neighbouring tiles - which share halo rows and the N-blocks of a tile,
  // which share the whole input tile - hit in the same L2.  Bijective for any grid size.
  int wid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = wid & 7, loc = wid >> 3;
    wid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int nb = wid % a.n_nblk;
  const int tflat = wid / a.n_nblk;
  int gi = 0;
  if (a.n_groups > 1 && tflat >= a.g[1].tile_begin) gi = 1;
  if (a.n_groups > 2 && tflat >= a.g[2].tile_begin) gi = 2;
  const MGroup& G = a.g[gi];
  const int tl = tflat - G.tile_begin;
  const int n0 = nb * NT;
  int b, ty0, tx0;
  long long flat0 = 0;
  const long long npix = (long long)G.B * G.H * G.W;
  if (a.flat) {
    flat0 = (long long)tl * (kTH * kTW);
    b = 0; ty0 = 0; tx0 = 0;
  } else {
    const int per_img = G.tiles_x * G.tiles_y;
    b = tl / per_img;
    const int t2 = tl % per_img;
    ty0 = (t2 / G.tiles_x) * kTH;
    tx0 = (t2 % G.tiles_x) * kTW;
  }

  f32x16_t acc[MW][NF];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][nf][i] = 0.f;

  for (int c0 = 0; c0 < a.cin16; c0 += kCK) {
    const int ck = (a.cin16 - c0) < kCK ? (a.cin16 - c0) : kCK;       // multiple of 16
    __syncthreads();   // every wave is done reading A_s / B_s of the previous chunk
    // ---- stage the halo tile of channels [c0, c0+64): f32 HBM -> 16-bit LDS -------------------------------------------
    if (a.planar) {
      // NCHW boundary input (7 or 21 frame planes): one scalar load per (pixel, channel); channels padded to 16 with zeros
      constexpr int NHP = HH * HWD;
      const View sv = G.src[0];
      const int cpad = a.cin16;                          // 16 or 32
      for (int idx = tid; idx < NHP * cpad; idx += 256) {
        const int hp = idx % NHP, c = idx / NHP;         // consecutive lanes = consecutive pixels of one plane (coalesced)
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int iy = ty0 + hy - PAD, ix = tx0 + hx - PAD;
        float v = 0.f;
        if (c < a.cin_total && iy >= 0 && iy < G.H && ix >= 0 && ix < G.W && !(a.dbg & 1))
          v = sv.p[(long long)b * sv.sb + (long long)iy * sv.sy + (long long)ix * sv.sx + (long long)c * sv.sc];
        const uint2 pk = cvt4<BF16>(make_float4(v, 0.f, 0.f, 0.f));
        A_s[hp * kLD + c] = (uint16_t)(pk.x & 0xffff);
      }
    } else if (!a.src16) {
      constexpr int NHP = HH * HWD;                    // halo pixels
      constexpr int ITERS = (NHP * 16 + 255) / 256;    // 16 channel-quads per pixel, 256 threads
      const int q = tid & 15;
      const int c = c0 + q * 4;
      const bool cok = (c < a.cin_total) && !(a.dbg & 1);
      int s_ = 0, cl = c;
      if (cl >= a.seg_c[0]) { cl -= a.seg_c[0]; s_ = 1; if (cl >= a.seg_c[1]) { cl -= a.seg_c[1]; s_ = 2; } }
      // The lane's source depends on its channel quad: G.src[s_] with a lane-dependent index is a VECTOR load from the
      // kernel-argument block; the three views' scalar fields are selected per lane instead.  The pixel loads are
      // unconditional (clamped coordinates, zeroed afterwards): predicated, each load of the unrolled group was its own
      // block and its own memory round trip.
      const int si = cok ? s_ : 0;
      const long long sp0 = sel3(si, uniform64((long long)G.src[0].p), uniform64((long long)G.src[1].p), uniform64((long long)G.src[2].p));
      const long long ssb = sel3(si, uniform64(G.src[0].sb), uniform64(G.src[1].sb), uniform64(G.src[2].sb));
      const long long ssy = sel3(si, uniform64(G.src[0].sy), uniform64(G.src[1].sy), uniform64(G.src[2].sy));
      const long long ssx = sel3(si, uniform64(G.src[0].sx), uniform64(G.src[1].sx), uniform64(G.src[2].sx));
      const long long sbase = sp0 + 4 * ((FLAT ? 0ll : (long long)b * ssb) + (cok ? cl : 0));      // byte address
      constexpr int UNR = 8;
#pragma unroll 1
      for (int it0 = 0; it0 < ITERS; it0 += UNR) {
        float4 v[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int hp = (tid >> 4) + (it0 + u) * 16;
          long long off;
          if (FLAT) {
            const long long p = flat0 + hp;
            ok[u] = cok && hp < NHP && p < npix;
            off = (p < npix ? p : npix - 1) * ssx;
          } else {
            const int hy = hp / HWD, hx = hp - hy * HWD;     // HWD is a compile-time constant (mul-shift)
            const int iy = ty0 + hy - PAD, ix = tx0 + hx - PAD;
            ok[u] = cok && hp < NHP && iy >= 0 && iy < G.H && ix >= 0 && ix < G.W;
            const int cy = iy < 0 ? 0 : (iy > G.H - 1 ? G.H - 1 : iy), cx = ix < 0 ? 0 : (ix > G.W - 1 ? G.W - 1 : ix);
            off = (long long)cy * ssy + (long long)cx * ssx;
          }
          v[u] = gld_f4(sbase + 4 * off);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (!ok[u]) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int hp = (tid >> 4) + (it0 + u) * 16;
          if (hp < NHP) *reinterpret_cast<uint2*>(A_s + hp * kLD + q * 4) = cvt4<BF16>(v[u]);
        }
      }
    } else {
      // sources already stored in the MFMA dtype: straight 16-byte copies (8 channels per lane, 8 lanes per pixel)
      constexpr int NHP = HH * HWD;
      constexpr int ITERS = (NHP * 8 + 255) / 256;
      const int q = tid & 7;
      const int c = c0 + q * 8;
      const bool cok = (c < a.cin_total) && !(a.dbg & 1);
      int s_ = 0, cl = c;
      if (cl >= a.seg_c[0]) { cl -= a.seg_c[0]; s_ = 1; if (cl >= a.seg_c[1]) { cl -= a.seg_c[1]; s_ = 2; } }
      const int si = cok ? s_ : 0;                     // scalar view fields selected per lane, unconditional loads (see above)
      const long long sp0 = sel3(si, uniform64((long long)G.src[0].p), uniform64((long long)G.src[1].p), uniform64((long long)G.src[2].p));
      const long long ssb = sel3(si, uniform64(G.src[0].sb), uniform64(G.src[1].sb), uniform64(G.src[2].sb));
      const long long ssy = sel3(si, uniform64(G.src[0].sy), uniform64(G.src[1].sy), uniform64(G.src[2].sy));
      const long long ssx = sel3(si, uniform64(G.src[0].sx), uniform64(G.src[1].sx), uniform64(G.src[2].sx));
      const long long sbase = sp0 + 2 * ((FLAT ? 0ll : (long long)b * ssb) + (cok ? cl : 0));      // byte address
      constexpr int UNR = 8;
#pragma unroll 1
      for (int it0 = 0; it0 < ITERS; it0 += UNR) {
        uint4 v[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int hp = (tid >> 3) + (it0 + u) * 32;
          long long off;
          if (FLAT) {
            const long long p = flat0 + hp;
            ok[u] = cok && hp < NHP && p < npix;
            off = (p < npix ? p : npix - 1) * ssx;
          } else {
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int iy = ty0 + hy - PAD, ix = tx0 + hx - PAD;
            ok[u] = cok && hp < NHP && iy >= 0 && iy < G.H && ix >= 0 && ix < G.W;
            const int cy = iy < 0 ? 0 : (iy > G.H - 1 ? G.H - 1 : iy), cx = ix < 0 ? 0 : (ix > G.W - 1 ? G.W - 1 : ix);
            off = (long long)cy * ssy + (long long)cx * ssx;
          }
          v[u] = gld_u4(sbase + 2 * off);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (!ok[u]) v[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int hp = (tid >> 3) + (it0 + u) * 32;
          if (hp < NHP) *reinterpret_cast<uint4*>(A_s + hp * kLD + q * 8) = v[u];
        }
      }
    }
    if (WD) {
      // ---- weights straight from L2/L1 into the MFMA B-operand registers: no LDS staging, NO barriers in the tap loop.
      // Lane (r,h) of N-fragment nf needs W[tap][n0+nf*32+r][c0+kk*16+h*8 .. +8]: one 16-byte load.  The flattened
      // (tap,kk) sequence is software-pipelined PD steps ahead through a register ring; waves free-run through the taps.
      constexpr int S = KS * KS * 4;
      constexpr int PD = NT == 128 ? 3 : 4;
      const long long wtap = (long long)a.cout_pad * a.cin_pad;
      const long long wrow32 = 32ll * a.cin_pad;
      const uint16_t* wl = a.w + ((long long)n0 + r) * a.cin_pad + c0 + h * 8;
      uint4 bq[PD][NF];
#pragma unroll
      for (int s0 = 0; s0 < PD && s0 < S; ++s0) {
        const int tap = s0 >> 2, kk = s0 & 3;
        if (kk * 16 < ck && !(a.dbg & 8)) {
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            bq[s0][nf] = *reinterpret_cast<const uint4*>(wl + tap * wtap + nf * wrow32 + kk * 16);
        }
      }
      __syncthreads();                     // A_s visible
#pragma unroll
      for (int s1 = 0; s1 < S; ++s1) {
        const int tap = s1 >> 2, kk = s1 & 3;
        const int ky = tap / KS, kx = tap - ky * KS;
        if (kk * 16 < ck && !(a.dbg & 2)) {
          const uint16_t* arow0 = A_s + ((MW * wave + ky) * HWD + r + kx) * kLD + h * 8;
          uint4 af[MW];
#pragma unroll
          for (int m = 0; m < MW; ++m) af[m] = *reinterpret_cast<const uint4*>(arow0 + m * HWD * kLD + kk * 16);
#pragma unroll
          for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[m][nf] = mfma<BF16>(af[m], bq[s1 % PD][nf], acc[m][nf]);
        }
        if (s1 + PD < S) {
          const int tap2 = (s1 + PD) >> 2, kk2 = (s1 + PD) & 3;
          if (kk2 * 16 < ck && !(a.dbg & 8)) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
              bq[s1 % PD][nf] = *reinterpret_cast<const uint4*>(wl + tap2 * wtap + nf * wrow32 + kk2 * 16);
          }
        }
      }
    } else {
    // ---- taps: weights global -> regs -> LDS, then MFMA ---------------------------------------------------------------
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;   // named (not an array): must stay in VGPRs
    // weights for chunk columns beyond ck are never read by the MFMA loop (kk*16 < ck), but the 16-byte loads must stay
    // inside the packed buffer: the packer pads cin_pad to a multiple of 64.
    const uint16_t* wbase = a.w + ((long long)n0 + (tid >> 3)) * a.cin_pad + c0 + (tid & 7) * 8;
    const long long wtap = (long long)a.cout_pad * a.cin_pad;   // halfwords per tap
    const long long wrow32 = 32ll * a.cin_pad;                   // i*256 threads = 32 more cout rows
#define FCVSR_FETCH_W(TAP)                                                                             \
  do {                                                                                                 \
    const uint16_t* wp_ = wbase + (TAP) * wtap;                                                        \
    if (a.dbg & 8) break;                                                                              \
    w0 = *reinterpret_cast<const uint4*>(wp_);                                                         \
    if (WLOADS > 1) w1 = *reinterpret_cast<const uint4*>(wp_ + wrow32);                                \
    if (WLOADS > 2) w2 = *reinterpret_cast<const uint4*>(wp_ + 2 * wrow32);                            \
    if (WLOADS > 3) w3 = *reinterpret_cast<const uint4*>(wp_ + 3 * wrow32);                            \
  } while (0)
    FCVSR_FETCH_W(0);
#pragma unroll 1
    for (int tap = 0; tap < KS * KS; ++tap) {
      if (tap > 0) __syncthreads();      // B_s of the previous tap fully consumed
      {
        uint16_t* bp = B_s + (tid >> 3) * kLD + (tid & 7) * 8;
        *reinterpret_cast<uint4*>(bp) = w0;
        if (WLOADS > 1) *reinterpret_cast<uint4*>(bp + 32 * kLD) = w1;
        if (WLOADS > 2) *reinterpret_cast<uint4*>(bp + 64 * kLD) = w2;
        if (WLOADS > 3) *reinterpret_cast<uint4*>(bp + 96 * kLD) = w3;
      }
      __syncthreads();                   // A_s (first tap) and B_s visible
      if (tap + 1 < KS * KS) FCVSR_FETCH_W(tap + 1);
      const int ky = tap / KS, kx = tap - ky * KS;
      const uint16_t* arow0 = A_s + ((MW * wave + ky) * HWD + r + kx) * kLD + h * 8;
      const uint16_t* brow = B_s + r * kLD + h * 8;
#pragma unroll
      for (int kk = 0; kk < kCK / 16; ++kk) {
        if (kk * 16 < ck && !(a.dbg & 2)) {
          uint4 af[MW], bf[NF];
#pragma unroll
          for (int m = 0; m < MW; ++m) af[m] = *reinterpret_cast<const uint4*>(arow0 + m * HWD * kLD + kk * 16);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) bf[nf] = *reinterpret_cast<const uint4*>(brow + nf * 32 * kLD + kk * 16);
#pragma unroll
          for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[m][nf] = mfma<BF16>(af[m], bf[nf], acc[m][nf]);
        }
      }
    }
    }
  }

  // ---- epilogue: accumulators -> per-wave LDS transpose -> 16-byte (4-channel) epilogue + stores ------------------------
  // A lane holds one cout column x 16 pixels; storing that directly is one dword per lane per instruction (store-issue
  // bound, ~1 TB/s).  Through LDS each lane gets 4 consecutive couts of one pixel: bias / residual loads and the output
  // store are 16 bytes per lane and a wave instruction covers whole 256-byte runs of a pixel's channels.
  float slope = a.slope;
  if (a.act == FCVSR_ACT_PRELU) slope = *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(a.slope_ptr));   // global, not flat: a flat_load turns every later counted lgkmcnt into lgkmcnt(0)
  EpiCtx e;
  e.act = a.act; e.slope = a.act == FCVSR_ACT_RELU ? 0.f : (a.act == FCVSR_ACT_NONE ? 1.f : slope); e.n_res = a.n_res; e.rs0 = a.rs[0]; e.rs1 = a.rs[1]; e.ps = a.ps; e.flat = a.flat;
  e.res0 = G.res[0]; e.res1 = G.res[1]; e.dst = G.dst; e.H = G.H; e.W = G.W; e.b = b; e.npix = npix;
  e.dst16 = a.dst16; e.dstbf = a.dstbf; e.cq4 = a.cout >> 2; e.sub2 = a.sub2;
  constexpr int EW = NT >= 64 ? 64 : 32;        // couts per pass
  constexpr int EROW = EW + 4;                  // padded row (floats): conflict-free b32 writes and b128 reads
  constexpr int QPR = EW / 4;                   // float4 per pixel row
  __syncthreads();                              // every wave is done with A_s / B_s
  float* E_s = reinterpret_cast<float*>(lds) + wave * (32 * EROW);
  const bool oct_path = a.dst16 && !a.ps && !a.gc_wmask && (a.cout % 8 == 0) && G.dst.sc == 1;
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    const int yo = MW * wave + m;
#pragma unroll
    for (int nh = 0; nh < NT / EW; ++nh) {
#pragma unroll
      for (int nf2 = 0; nf2 < EW / 32; ++nf2) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          E_s[((i & 3) + 8 * (i >> 2) + 4 * h) * EROW + nf2 * 32 + r] = acc[m][nh * (EW / 32) + nf2][i];
      }
      __builtin_amdgcn_wave_barrier();
      float gm = -INFINITY, gs = 0.f;                     // ContextBlock partials of this lane group (online softmax)
      float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oct_path) {
        constexpr int OPR = EW / 8;                       // lanes per pixel at 8 couts per lane
#pragma unroll
        for (int j = 0; j < 32 * OPR / 64; ++j) {
          const int idx = j * 64 + lane;
          const int co = idx % OPR, p = idx / OPR;
          const float4 va = *reinterpret_cast<const float4*>(E_s + p * EROW + co * 8);
          const float4 vb = *reinterpret_cast<const float4*>(E_s + p * EROW + co * 8 + 4);
          const int n = n0 + nh * EW + co * 8;
          if (n < a.cout && (!(a.dbg & 4) || va.x == 12345.678f))
            epilogue_oct<BF16>(e, va, vb, a.bias, n, ty0 + yo, tx0 + p, flat0 + yo * kTW + p);
          asm volatile("" ::: "memory");
        }
      } else
#pragma unroll
      for (int j = 0; j < 32 * QPR / 64; ++j) {
        const int idx = j * 64 + lane;
        const int cq = idx % QPR, p = idx / QPR;
        const float4 v = *reinterpret_cast<const float4*>(E_s + p * EROW + cq * 4);
        const int n = n0 + nh * EW + cq * 4;
        float4 xo = make_float4(0.f, 0.f, 0.f, 0.f);
        bool valid = false;
        if (!(a.dbg & 4) || v.x == 12345.678f)
          valid = epilogue_quad<BF16>(e, v, a.bias, a.cout, n, ty0 + yo, tx0 + p, flat0 + yo * kTW + p, &xo);
        if (a.gc_wmask) {
          // logit of pixel p = <r_p, wmask> summed over the QPR lanes that hold the pixel (ContextBlock.conv_mask, :676)
          float part = 0.f;
          if (valid) {
            const float4 wm = *reinterpret_cast<const float4*>(a.gc_wmask + n);
            part = xo.x * wm.x + (n + 1 < a.cout ? xo.y * wm.y : 0.f) + (n + 2 < a.cout ? xo.z * wm.z : 0.f) +
                   (n + 3 < a.cout ? xo.w * wm.w : 0.f);
          }
#pragma unroll
          for (int o = QPR / 2; o >= 1; o >>= 1) part += __shfl_xor(part, o);
          const bool pval = (ty0 + yo < G.H) && (tx0 + p < G.W);
          if (pval) {
            const float mn = fmaxf(gm, part);
            const float sc = expf(gm - mn);                 // gm = -inf on the first valid pixel -> 0
            const float ee = expf(part - mn);
            gs = gs * sc + ee;
            ga.x = ga.x * sc + ee * xo.x; ga.y = ga.y * sc + ee * xo.y; ga.z = ga.z * sc + ee * xo.z; ga.w = ga.w * sc + ee * xo.w;
            gm = mn;
          }
        }
        asm volatile("" ::: "memory");
      }
      if (a.gc_wmask && G.gc_partial) {
        // combine the 64/QPR lane groups of the wave, then lanes 0..QPR-1 write the wave's partial
#pragma unroll
        for (int o = QPR; o < 64; o <<= 1) {
          const float om = __shfl_xor(gm, o), os = __shfl_xor(gs, o);
          const float ox = __shfl_xor(ga.x, o), oy = __shfl_xor(ga.y, o), oz = __shfl_xor(ga.z, o), ow = __shfl_xor(ga.w, o);
          const float mn = fmaxf(gm, om);
          const float s1 = (gm == -INFINITY) ? 0.f : expf(gm - mn), s2 = (om == -INFINITY) ? 0.f : expf(om - mn);
          gs = gs * s1 + os * s2;
          ga.x = ga.x * s1 + ox * s2; ga.y = ga.y * s1 + oy * s2; ga.z = ga.z * s1 + oz * s2; ga.w = ga.w * s1 + ow * s2;
          gm = mn;
        }
        // per-wave partial -> LDS (behind the 4 transpose areas); combined across the 4 waves below
        float* gw = reinterpret_cast<float*>(lds) + 4 * 32 * EROW + wave * (EW + 4);
        if (lane < QPR) {
          *reinterpret_cast<float4*>(gw + lane * 4) = ga;
          if (lane == 0) { gw[EW] = gm; gw[EW + 1] = gs; }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (a.gc_wmask && G.gc_partial && NT <= 64 && MW == 1) {
    // one partial per workgroup: waves combined in fixed order (deterministic), written by wave 0
    __syncthreads();
    if (wave == 0 && lane < QPR) {
      const float* g0 = reinterpret_cast<const float*>(lds) + 4 * 32 * EROW;
      float m = -INFINITY;
#pragma unroll
      for (int w = 0; w < 4; ++w) m = fmaxf(m, g0[w * (EW + 4) + EW]);
      float sum = 0.f;
      float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float mw = g0[w * (EW + 4) + EW];
        const float sc = (mw == -INFINITY) ? 0.f : expf(mw - m);
        sum = fmaf(g0[w * (EW + 4) + EW + 1], sc, sum);
        const float4 v = *reinterpret_cast<const float4*>(g0 + w * (EW + 4) + lane * 4);
        acc4.x = fmaf(v.x, sc, acc4.x); acc4.y = fmaf(v.y, sc, acc4.y); acc4.z = fmaf(v.z, sc, acc4.z); acc4.w = fmaf(v.w, sc, acc4.w);
      }
      const int per_img = G.tiles_x * G.tiles_y;
      float* pp = G.gc_partial + ((long long)b * per_img + (tl % per_img)) * (a.cout + 2);
      const int n = lane * 4;
      if (n < a.cout) pp[n] = acc4.x;
      if (n + 1 < a.cout) pp[n + 1] = acc4.y;
      if (n + 2 < a.cout) pp[n + 2] = acc4.z;
      if (n + 3 < a.cout) pp[n + 3] = acc4.w;
      if (lane == 0) { pp[a.cout] = m; pp[a.cout + 1] = sum; }
    }
  }
}

// =====================================================================================================================
// Lean 3x3 kernel: the same algorithm as conv_mfma_kernel<.., KS=3, MW=1> specialised for the layers that carry the FLOPs
// (one dense NHWC source, stride 1, no pixel shuffle, channel-contiguous destination, cin % 64 == 0).  PMC showed the generic
// kernel is instruction-issue bound (1431 VALU + 1031 SALU instructions per wave for 72 MFMAs, SIMD issue 96 % busy):
// here every per-element index computation is hoisted - halo staging walks (row, col) incrementally with 32-bit offsets and
// constant LDS strides, the epilogue works on one tile row per wave with loop-invariant bias / channel offsets.
// =====================================================================================================================
template <bool BF16, int NT, bool SRC16, bool DST16>
__global__ __launch_bounds__(256, NT == 128 ? 2 : 4) void conv3_lean_kernel(MfmaArgs a) {
  constexpr int HWD = kTW + 2, NHP = 6 * HWD;          // 6 x 34 halo pixels
  constexpr int NF = NT / 32;
  constexpr int WLOADS = NT / 32;
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* A_s = lds;
  uint16_t* B_s = lds + NHP * kLD;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  int wid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = wid & 7, loc = wid >> 3;
    wid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int nb = wid % a.n_nblk;
  const int tflat = wid / a.n_nblk;
  int gi = 0;
  if (a.n_groups > 1 && tflat >= a.g[1].tile_begin) gi = 1;
  if (a.n_groups > 2 && tflat >= a.g[2].tile_begin) gi = 2;
  const MGroup& G = a.g[gi];
  const int tl = tflat - G.tile_begin;
  const int n0 = nb * NT;
  const int per_img = G.tiles_x * G.tiles_y;
  const int b = tl / per_img;
  const int t2 = tl - b * per_img;
  const int ty0 = (t2 / G.tiles_x) * 4, tx0 = (t2 % G.tiles_x) * kTW;
  const int H = G.H, W = G.W;

  f32x16_t acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;

  // ---- staging geometry (per thread, loop invariant) ---------------------------------------------------------------
  constexpr int EPL = SRC16 ? 8 : 4;                  // source elements per 16-byte load
  constexpr int LPP = 64 / EPL;                       // lanes per halo pixel (64 channels per chunk)
  constexpr int STEP = 256 / LPP;                     // halo pixels covered per iteration by the 256 threads
  constexpr int ITERS = (NHP + STEP - 1) / STEP;
  const int q = tid & (LPP - 1);
  const int p0 = tid / LPP;                           // < STEP <= 32 < HWD: first pixel lies in halo row 0
  const View sv = G.src[0];
  const int ssx = (int)sv.sx, ssy = (int)sv.sy;
  constexpr int ESZ = SRC16 ? 2 : 4;
  const char* sbase = reinterpret_cast<const char*>(sv.p) + ((long long)b * sv.sb) * ESZ;
  const int off0 = ((ty0 - 1) * ssy + (tx0 - 1 + p0) * ssx + q * EPL) * ESZ;      // byte offset of halo pixel p0
  uint16_t* a_dst = A_s + p0 * kLD + q * EPL;

  for (int c0 = 0; c0 < a.cin16; c0 += kCK) {
    const int ck = (a.cin16 - c0) < kCK ? (a.cin16 - c0) : kCK;
    __syncthreads();
    {
      int hy = 0, hx = p0, off = off0 + c0 * ESZ;
      const bool cok = (c0 + q * EPL) < a.cin_total;
      constexpr int SB = 7;                           // loads in flight per thread (register budget)
#pragma unroll
      for (int i0 = 0; i0 < ITERS; i0 += SB) {
        uint4 v[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int i = i0 + u;
          v[u] = make_uint4(0, 0, 0, 0);
          if (i < ITERS) {
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            const bool ok = cok && (p0 + i * STEP < NHP) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
            // unconditional (out-of-image / dead lanes read the image's first pixel and discard it): predicated loads of the
            // f32-source variants were issued one round trip after the other
            const uint4 ld = *reinterpret_cast<const uint4*>(sbase + (ok ? (unsigned)off : 0u));
            if (ok) v[u] = ld;
            hx += STEP;
            off += STEP * ssx * ESZ;
            if (hx >= HWD) { hx -= HWD; ++hy; off += (ssy - HWD * ssx) * ESZ; }
          }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int i = i0 + u;
          if (i < ITERS && p0 + i * STEP < NHP) {
            if (SRC16) {
              *reinterpret_cast<uint4*>(a_dst + i * STEP * kLD) = v[u];
            } else {
              const float4 f = __builtin_bit_cast(float4, v[u]);
              *reinterpret_cast<uint2*>(a_dst + i * STEP * kLD) = cvt4<BF16>(f);
            }
          }
        }
      }
    }
    // ---- taps: weights global -> regs -> LDS (one tap ahead), MFMA ---------------------------------------------------
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;
    const uint16_t* wbase = a.w + ((long long)n0 + (tid >> 3)) * a.cin_pad + c0 + (tid & 7) * 8;
    const long long wtap = (long long)a.cout_pad * a.cin_pad;
    const long long wrow32 = 32ll * a.cin_pad;
#define FCVSR_FETCH_WL(TAP)                                                       \
  do {                                                                            \
    const uint16_t* wp_ = wbase + (TAP) * wtap;                                   \
    w0 = *reinterpret_cast<const uint4*>(wp_);                                    \
    if (WLOADS > 1) w1 = *reinterpret_cast<const uint4*>(wp_ + wrow32);           \
    if (WLOADS > 2) w2 = *reinterpret_cast<const uint4*>(wp_ + 2 * wrow32);       \
    if (WLOADS > 3) w3 = *reinterpret_cast<const uint4*>(wp_ + 3 * wrow32);       \
  } while (0)
    FCVSR_FETCH_WL(0);
    uint16_t* bp = B_s + (tid >> 3) * kLD + (tid & 7) * 8;
    const uint16_t* arow = A_s + (wave * HWD + r) * kLD + h * 8;
    const uint16_t* brow = B_s + r * kLD + h * 8;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap > 0) __syncthreads();
      *reinterpret_cast<uint4*>(bp) = w0;
      if (WLOADS > 1) *reinterpret_cast<uint4*>(bp + 32 * kLD) = w1;
      if (WLOADS > 2) *reinterpret_cast<uint4*>(bp + 64 * kLD) = w2;
      if (WLOADS > 3) *reinterpret_cast<uint4*>(bp + 96 * kLD) = w3;
      __syncthreads();
      if (tap + 1 < 9) FCVSR_FETCH_WL(tap + 1);
      const int ky = tap / 3, kx = tap % 3;            // compile-time after unrolling
#pragma unroll
      for (int kk = 0; kk < kCK / 16; ++kk) {
        if (kk * 16 < ck) {
          const uint4 af = *reinterpret_cast<const uint4*>(arow + (ky * HWD + kx) * kLD + kk * 16);
          uint4 bf[NF];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) bf[nf] = *reinterpret_cast<const uint4*>(brow + nf * 32 * kLD + kk * 16);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) acc[nf] = mfma<BF16>(af, bf[nf], acc[nf]);
        }
      }
    }
  }

  // ---- epilogue: one tile row per wave ---------------------------------------------------------------------------------
  float slope = a.slope;
  if (a.act == FCVSR_ACT_PRELU) slope = *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(a.slope_ptr));   // global, not flat: a flat_load turns every later counted lgkmcnt into lgkmcnt(0)
  const int act = a.act;
  const float nsf = act == FCVSR_ACT_RELU ? 0.f : (act == FCVSR_ACT_NONE ? 1.f : slope);   // negative-side factor of the activation
  constexpr int EW = NT >= 64 ? 64 : 32;
  constexpr int EROW = EW + 4;
  __syncthreads();
  float* E_s = reinterpret_cast<float*>(lds) + wave * (32 * EROW);
  const int py = ty0 + wave;
  const bool rowok = py < H;
  const View dv = G.dst;
  const View r0v = G.res[0], r1v = G.res[1];
  const int dsx = (int)dv.sx, r0sx = (int)r0v.sx, r1sx = (int)r1v.sx;
  const long long drow = (long long)b * dv.sb + (long long)py * dv.sy;
  const long long r0off = (long long)b * r0v.sb + (long long)py * r0v.sy;
  const long long r1off = (long long)b * r1v.sb + (long long)py * r1v.sy;
  const bool r16 = a.res16 != 0;
  float gm = -INFINITY, gs = 0.f;
  float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int nh = 0; nh < NT / EW; ++nh) {
#pragma unroll
    for (int nf2 = 0; nf2 < EW / 32; ++nf2) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        E_s[((i & 3) + 8 * (i >> 2) + 4 * h) * EROW + nf2 * 32 + r] = acc[nh * (EW / 32) + nf2][i];
    }
    __builtin_amdgcn_wave_barrier();
    if (DST16) {
      // 8 couts per lane -> one 16-byte store; 8 lanes per pixel, 8 pixels per iteration
      constexpr int OPR = EW / 8;
      const int co = lane & (OPR - 1), psub = lane / OPR;
      const int n = n0 + nh * EW + co * 8;
      const bool nok = n < a.cout;
      float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
      if (a.bias && nok) { b0 = *reinterpret_cast<const float4*>(a.bias + n); b1 = *reinterpret_cast<const float4*>(a.bias + n + 4); }
      uint16_t* dp = reinterpret_cast<uint16_t*>(dv.p) + drow + n;
      const float* es = E_s + psub * EROW + co * 8;
      constexpr int PPI = 64 / OPR;                  // pixels per iteration
#pragma unroll
      for (int j = 0; j < 32 / PPI; ++j) {
        const int px = tx0 + j * PPI + psub;
        if (rowok && nok && px < W) {
          const float4 va = *reinterpret_cast<const float4*>(es + j * PPI * EROW);
          const float4 vb = *reinterpret_cast<const float4*>(es + j * PPI * EROW + 4);
          float x[8] = {va.x + b0.x, va.y + b0.y, va.z + b0.z, va.w + b0.w, vb.x + b1.x, vb.y + b1.y, vb.z + b1.z, vb.w + b1.w};
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = x[k] >= 0.f ? x[k] : x[k] * nsf;      // branch-free (hipcc does not unswitch on act)
          if (a.n_res > 0) {
            float rr[8];
            load_res<BF16, 8>(r0v.p, r0off + px * r0sx + n, r16, rr);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = fmaf(a.rs[0], rr[k], x[k]);
          }
          if (a.n_res > 1) {
            float rr[8];
            load_res<BF16, 8>(r1v.p, r1off + px * r1sx + n, r16, rr);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = fmaf(a.rs[1], rr[k], x[k]);
          }
          const uint2 lo = cvt4<BF16>(make_float4(x[0], x[1], x[2], x[3])), hi = cvt4<BF16>(make_float4(x[4], x[5], x[6], x[7]));
          *reinterpret_cast<uint4*>(dp + px * dsx) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      }
    } else {
      constexpr int QPR = EW / 4;
      const int cq = lane & (QPR - 1), psub = lane / QPR;
      const int n = n0 + nh * EW + cq * 4;
      const bool nok = n < a.cout;
      const bool full = n + 3 < a.cout;                 // skinny layers (conv_last0: cout 1 or 3) take the scalar path
      const bool vec = full && dv.sc == 1 && (a.n_res < 1 || r0v.sc == 1) && (a.n_res < 2 || r1v.sc == 1);
      float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias && nok) {
        if (full) b0 = *reinterpret_cast<const float4*>(a.bias + n);
        else { b0.x = a.bias[n]; if (n + 1 < a.cout) b0.y = a.bias[n + 1]; if (n + 2 < a.cout) b0.z = a.bias[n + 2]; }
      }
      float4 wm = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.gc_wmask && nok) wm = *reinterpret_cast<const float4*>(a.gc_wmask + n);
      float* dp = dv.p + drow + n;
      const float* es = E_s + psub * EROW + cq * 4;
      constexpr int PPI = 64 / QPR;
#pragma unroll
      for (int j = 0; j < 32 / PPI; ++j) {
        const int px = tx0 + j * PPI + psub;
        const bool pval = rowok && px < W;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pval && nok) {
          const float4 va = *reinterpret_cast<const float4*>(es + j * PPI * EROW);
          x = make_float4(va.x + b0.x, va.y + b0.y, va.z + b0.z, va.w + b0.w);
          x.x = x.x >= 0.f ? x.x : x.x * nsf; x.y = x.y >= 0.f ? x.y : x.y * nsf;
          x.z = x.z >= 0.f ? x.z : x.z * nsf; x.w = x.w >= 0.f ? x.w : x.w * nsf;
          if (vec) {
            if (a.n_res > 0) {
              float rr[4];
              load_res<BF16, 4>(r0v.p, r0off + px * r0sx + n, r16, rr);
              x.x = fmaf(a.rs[0], rr[0], x.x); x.y = fmaf(a.rs[0], rr[1], x.y); x.z = fmaf(a.rs[0], rr[2], x.z); x.w = fmaf(a.rs[0], rr[3], x.w);
            }
            if (a.n_res > 1) {
              float rr[4];
              load_res<BF16, 4>(r1v.p, r1off + px * r1sx + n, r16, rr);
              x.x = fmaf(a.rs[1], rr[0], x.x); x.y = fmaf(a.rs[1], rr[1], x.y); x.z = fmaf(a.rs[1], rr[2], x.z); x.w = fmaf(a.rs[1], rr[3], x.w);
            }
            if (a.gc16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(dv.p) + drow + n + px * dsx) = cvt4<BF16>(x);
            else *reinterpret_cast<float4*>(dp + px * dsx) = x;
          } else {
            float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (n + k < a.cout) {
                if (a.n_res > 0) xs[k] = fmaf(a.rs[0], r0v.p[r0off + px * r0sx + (long long)(n + k) * r0v.sc], xs[k]);
                if (a.n_res > 1) xs[k] = fmaf(a.rs[1], r1v.p[r1off + px * r1sx + (long long)(n + k) * r1v.sc], xs[k]);
                dv.p[drow + px * dsx + (long long)(n + k) * dv.sc] = xs[k];
              } else {
                xs[k] = 0.f;
              }
            }
            x = make_float4(xs[0], xs[1], xs[2], xs[3]);
          }
        }
        if (a.gc_wmask) {
          float part = x.x * wm.x + x.y * wm.y + x.z * wm.z + x.w * wm.w;     // x == 0 on invalid lanes
#pragma unroll
          for (int o = QPR / 2; o >= 1; o >>= 1) part += __shfl_xor(part, o);
          if (pval) {
            const float mn = fmaxf(gm, part);
            const float sc = expf(gm - mn), ee = expf(part - mn);
            gs = gs * sc + ee;
            ga.x = ga.x * sc + ee * x.x; ga.y = ga.y * sc + ee * x.y; ga.z = ga.z * sc + ee * x.z; ga.w = ga.w * sc + ee * x.w;
            gm = mn;
          }
        }
      }
      if (a.gc_wmask && G.gc_partial) {
#pragma unroll
        for (int o = QPR; o < 64; o <<= 1) {
          const float om = __shfl_xor(gm, o), os = __shfl_xor(gs, o);
          const float ox = __shfl_xor(ga.x, o), oy = __shfl_xor(ga.y, o), oz = __shfl_xor(ga.z, o), ow = __shfl_xor(ga.w, o);
          const float mn = fmaxf(gm, om);
          const float s1 = (gm == -INFINITY) ? 0.f : expf(gm - mn), s2 = (om == -INFINITY) ? 0.f : expf(om - mn);
          gs = gs * s1 + os * s2;
          ga.x = ga.x * s1 + ox * s2; ga.y = ga.y * s1 + oy * s2; ga.z = ga.z * s1 + oz * s2; ga.w = ga.w * s1 + ow * s2;
          gm = mn;
        }
        float* gw = reinterpret_cast<float*>(lds) + 4 * 32 * EROW + wave * (EW + 4);
        if (lane < QPR) {
          *reinterpret_cast<float4*>(gw + lane * 4) = ga;
          if (lane == 0) { gw[EW] = gm; gw[EW + 1] = gs; }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!DST16 && a.gc_wmask && G.gc_partial && NT <= 64) {
    constexpr int QPR = EW / 4;
    __syncthreads();
    if (wave == 0 && lane < QPR) {
      const float* g0 = reinterpret_cast<const float*>(lds) + 4 * 32 * EROW;
      float m = -INFINITY;
#pragma unroll
      for (int w = 0; w < 4; ++w) m = fmaxf(m, g0[w * (EW + 4) + EW]);
      float sum = 0.f;
      float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float mw = g0[w * (EW + 4) + EW];
        const float sc = (mw == -INFINITY) ? 0.f : expf(mw - m);
        sum = fmaf(g0[w * (EW + 4) + EW + 1], sc, sum);
        const float4 v = *reinterpret_cast<const float4*>(g0 + w * (EW + 4) + lane * 4);
        acc4.x = fmaf(v.x, sc, acc4.x); acc4.y = fmaf(v.y, sc, acc4.y); acc4.z = fmaf(v.z, sc, acc4.z); acc4.w = fmaf(v.w, sc, acc4.w);
      }
      float* pp = G.gc_partial + ((long long)b * per_img + t2) * (a.cout + 2);
      const int n = lane * 4;
      if (n < a.cout) pp[n] = acc4.x;
      if (n + 1 < a.cout) pp[n + 1] = acc4.y;
      if (n + 2 < a.cout) pp[n + 2] = acc4.z;
      if (n + 3 < a.cout) pp[n + 3] = acc4.w;
      if (lane == 0) { pp[a.cout] = m; pp[a.cout + 1] = sum; }
    }
  }
}

// Lean 1x1 kernel: flat list of pixels (128 per workgroup), up to 3 concatenated sources whose channel counts are
// multiples of 64 (a 64-channel chunk never straddles two sources), optional pixel-shuffle destination (no residuals then).
template <bool BF16, int NT, bool SRC16, bool DST16, bool PS>
__global__ __launch_bounds__(256, 4) void conv1_lean_kernel(MfmaArgs a) {
  constexpr int NPX = 128;
  constexpr int NF = NT / 32;
  constexpr int WLOADS = NT / 32;
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* A_s = lds;
  uint16_t* B_s = lds + NPX * kLD;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int wid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = wid & 7, loc = wid >> 3;
    wid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int nb = wid % a.n_nblk;
  const int tflat = wid / a.n_nblk;
  int gi = 0;
  if (a.n_groups > 1 && tflat >= a.g[1].tile_begin) gi = 1;
  if (a.n_groups > 2 && tflat >= a.g[2].tile_begin) gi = 2;
  const MGroup& G = a.g[gi];
  const int tl = tflat - G.tile_begin;
  const int n0 = nb * NT;
  const int npix = G.B * G.H * G.W;                   // host guarantees < 2^29
  const int flat0 = tl * NPX;

  f32x16_t acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;

  constexpr int EPL = SRC16 ? 8 : 4;
  constexpr int LPP = 64 / EPL;
  constexpr int STEP = 256 / LPP;
  constexpr int ITERS = NPX / STEP;
  constexpr int ESZ = SRC16 ? 2 : 4;
  const int q = tid & (LPP - 1);
  const int p0 = tid / LPP;
  uint16_t* a_dst = A_s + p0 * kLD + q * EPL;

  for (int c0 = 0; c0 < a.cin16; c0 += kCK) {
    // source segment of this chunk (uniform): segments are multiples of 64 channels
    int sidx = 0, cl = c0;
    if (cl >= a.seg_c[0]) { cl -= a.seg_c[0]; sidx = 1; if (cl >= a.seg_c[1]) { cl -= a.seg_c[1]; sidx = 2; } }
    const View sv = G.src[sidx];
    const char* sbase = reinterpret_cast<const char*>(sv.p) + (cl + q * EPL) * ESZ;
    const int ssx = (int)sv.sx;
    __syncthreads();
    {
      uint4 v[ITERS];
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        const int pix = flat0 + p0 + i * STEP;
        const long long pc = pix < npix ? pix : npix - 1;            // clamped, not predicated (rows past the end are never stored)
        v[i] = *reinterpret_cast<const uint4*>(sbase + (size_t)((unsigned)pc * (unsigned)ssx) * ESZ);
      }
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        if (SRC16) {
          *reinterpret_cast<uint4*>(a_dst + i * STEP * kLD) = v[i];
        } else {
          *reinterpret_cast<uint2*>(a_dst + i * STEP * kLD) = cvt4<BF16>(__builtin_bit_cast(float4, v[i]));
        }
      }
    }
    {
      const uint16_t* wp_ = a.w + ((long long)n0 + (tid >> 3)) * a.cin_pad + c0 + (tid & 7) * 8;
      const long long wrow32 = 32ll * a.cin_pad;
      uint16_t* bp = B_s + (tid >> 3) * kLD + (tid & 7) * 8;
      const uint4 w0 = *reinterpret_cast<const uint4*>(wp_);
      uint4 w1 = w0;
      if (WLOADS > 1) w1 = *reinterpret_cast<const uint4*>(wp_ + wrow32);
      *reinterpret_cast<uint4*>(bp) = w0;
      if (WLOADS > 1) *reinterpret_cast<uint4*>(bp + 32 * kLD) = w1;
    }
    __syncthreads();
    const uint16_t* arow = A_s + (wave * 32 + r) * kLD + h * 8;
    const uint16_t* brow = B_s + r * kLD + h * 8;
#pragma unroll
    for (int kk = 0; kk < kCK / 16; ++kk) {
      const uint4 af = *reinterpret_cast<const uint4*>(arow + kk * 16);
      uint4 bf[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) bf[nf] = *reinterpret_cast<const uint4*>(brow + nf * 32 * kLD + kk * 16);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[nf] = mfma<BF16>(af, bf[nf], acc[nf]);
    }
  }

  float slope = a.slope;
  if (a.act == FCVSR_ACT_PRELU) slope = *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(a.slope_ptr));   // global, not flat: a flat_load turns every later counted lgkmcnt into lgkmcnt(0)
  const int act = a.act;
  const float nsf = act == FCVSR_ACT_RELU ? 0.f : (act == FCVSR_ACT_NONE ? 1.f : slope);   // negative-side factor of the activation
  constexpr int EW = NT >= 64 ? 64 : 32;
  constexpr int EROW = EW + 4;
  __syncthreads();
  float* E_s = reinterpret_cast<float*>(lds) + wave * (32 * EROW);
  const int rowpix = flat0 + wave * 32;
  const View dv = G.dst;
  const View r0v = G.res[0], r1v = G.res[1];
  const int dsx = (int)dv.sx, r0sx = (int)r0v.sx, r1sx = (int)r1v.sx;
  const int cq4 = a.cout >> 2;
#pragma unroll
  for (int nf2 = 0; nf2 < EW / 32; ++nf2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) E_s[((i & 3) + 8 * (i >> 2) + 4 * h) * EROW + nf2 * 32 + r] = acc[nf2][i];
  }
  __builtin_amdgcn_wave_barrier();
  constexpr int VPL = DST16 ? 8 : 4;                  // couts per lane
  constexpr int LPR = EW / VPL;                       // lanes per pixel
  constexpr int PPI = 64 / LPR;
  const int co = lane & (LPR - 1), psub = lane / LPR;
  const int n = n0 + co * VPL;
  const bool nok = n < a.cout;
  float bb[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) bb[k] = 0.f;
  if (a.bias && nok) {
#pragma unroll
    for (int k = 0; k < VPL; k += 4) {
      const float4 t = *reinterpret_cast<const float4*>(a.bias + n + k);
      bb[k] = t.x; bb[k + 1] = t.y; bb[k + 2] = t.z; bb[k + 3] = t.w;
    }
  }
  // pixel-shuffle: couts are sub-pixel-major, n = sp*(cout/4) + c
  const int sp = PS ? n / cq4 : 0;
  const int nn = PS ? n - sp * cq4 : n;
  const float* es = E_s + psub * EROW + co * VPL;
#pragma unroll
  for (int j = 0; j < 32 / PPI; ++j) {
    const int pix = rowpix + j * PPI + psub;
    if (nok && pix < npix) {
      float x[VPL];
#pragma unroll
      for (int k = 0; k < VPL; k += 4) {
        const float4 t = *reinterpret_cast<const float4*>(es + j * PPI * EROW + k);
        x[k] = t.x + bb[k]; x[k + 1] = t.y + bb[k + 1]; x[k + 2] = t.z + bb[k + 2]; x[k + 3] = t.w + bb[k + 3];
      }
#pragma unroll
      for (int k = 0; k < VPL; ++k) x[k] = x[k] >= 0.f ? x[k] : x[k] * nsf;
      if (!PS) {
        if (a.n_res > 0) {
#pragma unroll
          for (int k = 0; k < VPL; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(r0v.p + (size_t)((unsigned)pix * (unsigned)r0sx) + n + k);
            x[k] = fmaf(a.rs[0], t.x, x[k]); x[k + 1] = fmaf(a.rs[0], t.y, x[k + 1]);
            x[k + 2] = fmaf(a.rs[0], t.z, x[k + 2]); x[k + 3] = fmaf(a.rs[0], t.w, x[k + 3]);
          }
        }
        if (a.n_res > 1) {
#pragma unroll
          for (int k = 0; k < VPL; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(r1v.p + (size_t)((unsigned)pix * (unsigned)r1sx) + n + k);
            x[k] = fmaf(a.rs[1], t.x, x[k]); x[k + 1] = fmaf(a.rs[1], t.y, x[k + 1]);
            x[k + 2] = fmaf(a.rs[1], t.z, x[k + 2]); x[k + 3] = fmaf(a.rs[1], t.w, x[k + 3]);
          }
        }
      }
      size_t o;
      if (PS) {
        const int qx = pix % G.W, t1 = pix / G.W;
        const int qy = t1 % G.H, qb = t1 / G.H;
        o = (size_t)qb * dv.sb + (size_t)(2 * qy + (sp >> 1)) * dv.sy + (size_t)(2 * qx + (sp & 1)) * dsx + nn;
      } else {
        o = (size_t)((unsigned)pix * (unsigned)dsx) + n;
      }
      if (DST16) {
        const uint2 lo = cvt4<BF16>(make_float4(x[0], x[1], x[2], x[3]));
        const uint2 hi = cvt4<BF16>(make_float4(x[VPL - 4], x[VPL - 3], x[VPL - 2], x[VPL - 1]));
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(dv.p) + o) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      } else {
        *reinterpret_cast<float4*>(dv.p + o) = make_float4(x[0], x[1], x[2], x[3]);
      }
    }
  }
}

template <bool BF16, int NT, bool SRC16, bool DST16, bool PS>
static hipError_t launch_lean1(const MfmaArgs& a, int total_tiles, hipStream_t st) {
  size_t lds = ((size_t)128 * kLD + (size_t)NT * kLD) * sizeof(uint16_t);
  const size_t epi = 4ull * 32 * ((NT >= 64 ? 64 : 32) + 4) * sizeof(float);
  if (lds < epi) lds = epi;
  hipLaunchKernelGGL((conv1_lean_kernel<BF16, NT, SRC16, DST16, PS>), dim3(total_tiles * a.n_nblk), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <bool BF16>
static hipError_t dispatch_lean1(const MfmaArgs& a, int nt, int total_tiles, hipStream_t st) {
#define FCVSR_L1(NTV, PSV)                                                                                      \
  (a.src16 ? (a.dst16 ? launch_lean1<BF16, NTV, true, true, PSV>(a, total_tiles, st)                              \
                      : launch_lean1<BF16, NTV, true, false, PSV>(a, total_tiles, st))                            \
           : (a.dst16 ? launch_lean1<BF16, NTV, false, true, PSV>(a, total_tiles, st)                             \
                      : launch_lean1<BF16, NTV, false, false, PSV>(a, total_tiles, st)))
  if (a.ps) return nt == 64 ? FCVSR_L1(64, true) : FCVSR_L1(32, true);
  return nt == 64 ? FCVSR_L1(64, false) : FCVSR_L1(32, false);
#undef FCVSR_L1
}

template <bool BF16, int NT, bool SRC16, bool DST16>
static hipError_t launch_lean(const MfmaArgs& a, int total_tiles, hipStream_t st) {
  size_t lds = ((size_t)6 * (kTW + 2) * kLD + (size_t)NT * kLD) * sizeof(uint16_t);
  const size_t epi = (4ull * 32 + 4) * ((NT >= 64 ? 64 : 32) + 4) * sizeof(float);
  if (lds < epi) lds = epi;
  static DevOnce attr;
  hipError_t e = once_per_device(attr, [&] {
    return hipFuncSetAttribute((const void*)conv3_lean_kernel<BF16, NT, SRC16, DST16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((conv3_lean_kernel<BF16, NT, SRC16, DST16>), dim3(total_tiles * a.n_nblk), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <bool BF16>
static hipError_t dispatch_lean(const MfmaArgs& a, int nt, int total_tiles, hipStream_t st) {
#define FCVSR_LEAN(NTV)                                                                                       \
  (a.src16 ? (a.dst16 ? launch_lean<BF16, NTV, true, true>(a, total_tiles, st)                                  \
                      : launch_lean<BF16, NTV, true, false>(a, total_tiles, st))                                \
           : (a.dst16 ? launch_lean<BF16, NTV, false, true>(a, total_tiles, st)                                 \
                      : launch_lean<BF16, NTV, false, false>(a, total_tiles, st)))
  if (nt == 64) return FCVSR_LEAN(64);     // (the 128-cout tile measured 0.65x and spilled: no longer built)
  return FCVSR_LEAN(32);
#undef FCVSR_LEAN
}

template <bool BF16, int NT, int KS, int MW, bool WD>
static hipError_t launch_mfma(const MfmaArgs& a, int total_tiles, hipStream_t st) {
  constexpr int PAD = KS / 2;
  constexpr int kTH = 4 * MW;
  size_t lds = ((size_t)(kTH + 2 * PAD) * (kTW + 2 * PAD) * kLD + (size_t)NT * kLD) * sizeof(uint16_t);
  const size_t epi = (4ull * 32 + 4) * ((NT >= 64 ? 64 : 32) + 4) * sizeof(float);   // epilogue transpose + GC partial areas
  if (lds < epi) lds = epi;
  static DevOnce attr;
  hipError_t e = once_per_device(attr, [&] {
    return hipFuncSetAttribute((const void*)conv_mfma_kernel<BF16, NT, KS, MW, WD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((conv_mfma_kernel<BF16, NT, KS, MW, WD>), dim3(total_tiles * a.n_nblk), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <bool BF16, int MW, bool WD>
static hipError_t dispatch2(const MfmaArgs& a, int nt, int ks, int total_tiles, hipStream_t st) {
  if (ks == 3) {
    if (nt == 64) return launch_mfma<BF16, 64, 3, MW, WD>(a, total_tiles, st);
    return launch_mfma<BF16, 32, 3, MW, WD>(a, total_tiles, st);
  }
  if (nt == 64) return launch_mfma<BF16, 64, 1, MW, WD>(a, total_tiles, st);
  return launch_mfma<BF16, 32, 1, MW, WD>(a, total_tiles, st);
}

template <bool BF16>
static hipError_t dispatch(const MfmaArgs& a, int nt, int ks, int mw, int wd, int total_tiles, hipStream_t st) {
  (void)mw;      // (8-row tiles - MW = 2 - measured 0.85-1.0x and their 128-cout instances spilled 176 scratch operations: not built)
  return wd ? dispatch2<BF16, 1, true>(a, nt, ks, total_tiles, st) : dispatch2<BF16, 1, false>(a, nt, ks, total_tiles, st);
}

}  // namespace fcvsr

using namespace fcvsr;

// diagnostic stamp buffer of the resident-weight kernel (FCVSR_RES_STAMPS=1): [wave 8][phase 64][slot 8] u64
static void* g_res_stamps = nullptr;
static const size_t kResStampBytes = 8 * 64 * 8 * sizeof(unsigned long long);
extern "C" int fcvsr_debug_res_stamps(void* host_out, size_t bytes) {
  if (!g_res_stamps || bytes > kResStampBytes) return FCVSR_E_ARG;
  return (int)hipMemcpy(host_out, g_res_stamps, bytes, hipMemcpyDeviceToHost);
}

// name of the kernel the last fcvsr_conv2d_mfma call of this thread launched (bench.py groups its per-launch timings by it)
static thread_local char g_last_kernel[96] = "";
extern "C" const char* fcvsr_last_conv_kernel(void) { return g_last_kernel; }
#define FCVSR_NOTE_KERNEL(...) snprintf(g_last_kernel, sizeof(g_last_kernel), __VA_ARGS__)
static const char* tf(int v) { return v ? "true" : "false"; }

// channel-contiguous res/dst views are accessed 16 bytes at a time
static bool vec_view_ok(const fcvsr_view& v) {
  const int g = v.dtype == FCVSR_F32 ? 4 : 8;   // 16-byte granules for f32 quads, 8-byte for 16-bit quads (keep 16 for safety)
  return v.sc != 1 || v.c < 4 || (v.sx % 4 == 0 && v.sy % 4 == 0 && v.sb % 4 == 0 && ((uintptr_t)v.ptr % (g == 4 ? 16 : 8)) == 0);
}

// sources: f32 (converted while staging, 4 channels = 16 bytes per lane) or already in the MFMA dtype (8 channels per lane)
static bool src_ok(const fcvsr_view& v, int mma_dtype) {
  if (!v.ptr || v.sc != 1 || ((uintptr_t)v.ptr % 16) != 0) return false;
  const int g = v.dtype == FCVSR_F32 ? 4 : 8;
  if (v.dtype != FCVSR_F32 && v.dtype != mma_dtype) return false;
  return v.c % g == 0 && v.sx % g == 0 && v.sy % g == 0 && v.sb % g == 0;
}

extern "C" int fcvsr_conv2d_mfma(const fcvsr_conv_desc* descs, int n_groups, int mma_dtype, void* stream) {
  FCVSR_CHECK_ARG(descs != nullptr && n_groups >= 1 && n_groups <= 3, "1..3 problem groups");
  FCVSR_CHECK_ARG(mma_dtype == FCVSR_BF16 || mma_dtype == FCVSR_F16, "mma_dtype must be BF16 or F16");
  const fcvsr_conv_desc& d0 = descs[0];
  FCVSR_CHECK_ARG(d0.kh == d0.kw && (d0.kh == 1 || d0.kh == 3) && d0.pad == d0.kh / 2 &&
                      (d0.stride == 1 || (d0.stride == 2 && d0.kh == 3 && !d0.pixel_shuffle)),
                  "MFMA path: 1x1 or 3x3 stride 1, or 3x3 stride 2, same padding");
  FCVSR_CHECK_ARG(d0.n_src >= 1 && d0.n_src <= 3 && d0.n_res >= 0 && d0.n_res <= 2 && d0.cout > 0, "bad descriptor");
  FCVSR_CHECK_ARG(d0.weight != nullptr && d0.cout_pad % 128 == 0 && d0.cout_pad >= d0.cout, "weight must be MFMA-packed");
  FCVSR_CHECK_ARG(!(d0.act == FCVSR_ACT_PRELU) || d0.slope_ptr != nullptr, "PReLU needs slope_ptr");
  FCVSR_CHECK_ARG(!d0.pixel_shuffle || d0.cout % 16 == 0, "pixel_shuffle needs cout%16==0 (sub-pixel-major packing)");
  MfmaArgs a;
  a.src16 = d0.src[0].dtype != FCVSR_F32;
  a.res16 = (d0.n_res > 0 && d0.res[0].dtype != FCVSR_F32) ? 1 : 0;
  a.sub2 = d0.stride == 2;
  a.planar = (d0.n_src == 1 && d0.src[0].sc != 1) ? 1 : 0;
  FCVSR_CHECK_ARG(!a.planar || (d0.src[0].dtype == FCVSR_F32 && d0.src[0].c <= 32 && d0.kh == 3),
                  "planar (channel-strided) source: one f32 source with <= 32 channels, 3x3");
  a.dst16 = d0.dst.dtype != FCVSR_F32;
  a.gc16 = 0;
  a.dstbf = d0.dst.dtype == FCVSR_BF16;
  const bool dst_native = d0.dst.dtype == FCVSR_F32 || d0.dst.dtype == mma_dtype;   // lean kernels store f32 / MFMA dtype only
  a.n_groups = n_groups;
  a.n_src = d0.n_src;
  a.n_res = d0.n_res;
  int cin = 0;
  for (int s = 0; s < 3; ++s) {
    a.seg_c[s] = s < d0.n_src ? d0.src[s].c : (1 << 30);
    if (s < d0.n_src) cin += d0.src[s].c;
  }
  a.cin_total = cin;
  a.cin16 = (cin + 15) / 16 * 16;
  a.cin_pad = (cin + 63) / 64 * 64;   // packer pads cin to a multiple of 64 (16-byte weight loads stay in bounds)
  a.cout = d0.cout;
  a.cout_pad = d0.cout_pad;
  // N tile: measured faster with <= 64 couts per workgroup (register pressure of 128-cout accumulators costs more than
  // re-staging the input tile for the second N-block); FCVSR_MFMA_NTMAX=128 restores the wide tile for experiments.
  int nt = d0.cout > 32 ? 64 : 32;
  a.n_nblk = (d0.cout + nt - 1) / nt;
  a.w = (const uint16_t*)d0.weight;
  a.bias = d0.bias;
  a.act = d0.act;
  a.slope = d0.slope;
  a.slope_ptr = d0.slope_ptr;
  a.rs[0] = d0.res_scale[0];
  a.rs[1] = d0.res_scale[1];
  a.ps = d0.pixel_shuffle;
  a.flat = d0.kh == 1 ? 1 : 0;
  a.gc_wmask = d0.gc_wmask;
  FCVSR_CHECK_ARG(d0.gc_wmask == nullptr || (d0.kh == 3 && d0.stride == 1 && d0.cout <= 64 && !d0.pixel_shuffle &&
                                             ((uintptr_t)d0.gc_wmask % 16) == 0),
                  "ContextBlock fusion: 3x3 stride-1 layer with cout <= 64");
  {
    const char* dbg = getenv("FCVSR_MFMA_DBG");
    a.dbg = dbg ? atoi(dbg) : 0;
  }
  // tile rows per workgroup: 8 (2 per wave) or 4 (1 per wave: half the LDS -> more co-resident workgroups in
  // different phases).  FCVSR_MFMA_MW overrides for experiments.
  const int mw = 1;   // measured: 4-row tiles (more co-resident workgroups in different phases) win for 3x3 and 1x1 alike
  int wd = 0;     // weights straight from L2 into the B-operand registers (no LDS staging / tap barriers)
  {
    const char* e = getenv("FCVSR_MFMA_WD");
    if (e) wd = atoi(e) ? 1 : 0;
  }
  const int kTH = 4 * mw;
  int tiles = 0;
  for (int g = 0; g < n_groups; ++g) {
    const fcvsr_conv_desc& d = descs[g];
    FCVSR_CHECK_ARG(d.kh == d0.kh && d.kw == d0.kw && d.stride == d0.stride && d.n_src == d0.n_src && d.n_res == d0.n_res &&
                        d.cout == d0.cout && d.weight == d0.weight && d.bias == d0.bias && d.act == d0.act &&
                        d.pixel_shuffle == d0.pixel_shuffle,
                    "groups must share weights and epilogue");
    FCVSR_CHECK_ARG(d.B > 0 && d.H > 0 && d.W > 0, "empty problem");
    MGroup& G = a.g[g];
    for (int s = 0; s < d.n_src; ++s) {
      FCVSR_CHECK_ARG((a.planar ? (d.src[s].ptr != nullptr && d.src[s].sc != 1) : src_ok(d.src[s], mma_dtype)) &&
                          d.src[s].c == d0.src[s].c && d.src[s].dtype == d0.src[0].dtype,
                      "src: f32 or MFMA dtype (all alike), channel-contiguous, 16-byte aligned, c%4==0 (f32) / c%8==0 (16-bit)");
      G.src[s] = to_view(d.src[s]);
      if (a.flat)
        FCVSR_CHECK_ARG(d.src[s].sy == d.src[s].sx * d.W && d.src[s].sb == d.src[s].sy * d.H, "1x1 needs uniformly strided pixels");
    }
    for (int q = 0; q < d.n_res; ++q) {
      FCVSR_CHECK_ARG(d.res[q].ptr && (d.res[q].dtype == FCVSR_F32 || d.res[q].dtype == mma_dtype) &&
                          d.res[q].dtype == d0.res[0].dtype && vec_view_ok(d.res[q]),
                      "res must be f32 or the MFMA dtype (all alike), vector-aligned");
      G.res[q] = to_view(d.res[q]);
      if (a.flat)
        FCVSR_CHECK_ARG(d.res[q].sy == d.res[q].sx * d.W && d.res[q].sb == d.res[q].sy * d.H, "1x1 needs uniformly strided res");
    }
    FCVSR_CHECK_ARG(d.dst.ptr && (d.dst.dtype == FCVSR_F32 || d.dst.dtype == FCVSR_BF16 || d.dst.dtype == FCVSR_F16) &&
                        d.dst.dtype == d0.dst.dtype && vec_view_ok(d.dst), "dst must be f32 / bf16 / f16, vector-aligned");
    FCVSR_CHECK_ARG(d.bias == nullptr || ((uintptr_t)d.bias % 16) == 0, "bias must be 16-byte aligned");
    G.dst = to_view(d.dst);
    if (a.flat && !d.pixel_shuffle)
      FCVSR_CHECK_ARG(d.dst.sy == d.dst.sx * d.W && d.dst.sb == d.dst.sy * d.H, "1x1 needs uniformly strided dst");
    G.gc_partial = d.gc_partial;
    FCVSR_CHECK_ARG((d.gc_wmask == nullptr) == (d.gc_partial == nullptr) && d.gc_wmask == d0.gc_wmask, "gc fields: all groups alike");
    G.B = d.B; G.H = d.H; G.W = d.W;
    G.tile_begin = tiles;
    if (a.flat) {
      G.tiles_x = 1; G.tiles_y = 1;
      tiles += cdiv((long long)d.B * d.H * d.W, kTH * kTW);
    } else {
      G.tiles_x = cdiv(d.W, kTW);
      G.tiles_y = cdiv(d.H, kTH);
      tiles += d.B * G.tiles_x * G.tiles_y;
    }
  }
  for (int g = n_groups; g < 3; ++g) a.g[g] = a.g[0];
  hipStream_t st = (hipStream_t)stream;
  // lean fast path: 3x3 stride 1, one dense source, cin multiple of 64, plain channel-contiguous destination and residuals
  bool lean = dst_native && d0.kh == 3 && d0.stride == 1 && mw == 1 && !wd && d0.n_src == 1 && !a.planar && !a.ps && (cin % 64 == 0) &&
              (!a.dst16 || d0.cout % 8 == 0) && (d0.gc_wmask == nullptr || d0.cout % 4 == 0);
  for (int g = 0; g < n_groups && lean; ++g) {
    const fcvsr_conv_desc& d = descs[g];
    long long ext = (long long)d.B * d.H * d.W * (long long)(d.src[0].sx > d.dst.sx ? d.src[0].sx : d.dst.sx);
    if (d.dst.sc != 1) ext = (long long)d.B * d.dst.sb;                 // strided (e.g. NCHW) destination
    lean = lean && (!a.dst16 || d.dst.sc == 1) && ext < (1ll << 29);
    for (int q = 0; q < d.n_res; ++q)
      lean = lean && (d.res[q].dtype == FCVSR_F32 || d.res[q].sc == 1) && (!a.dst16 || d.res[q].sc == 1);
  }
  if (d0.gc_wmask != nullptr && a.dst16) {
    // ContextBlock partials come out of the 4-couts-per-lane epilogue: run that variant and let it store 16-bit values
    bool ok16 = lean && dst_native && d0.n_res == 0 && d0.cout % 4 == 0;
    for (int g = 0; g < n_groups; ++g) ok16 = ok16 && descs[g].dst.sc == 1;
    FCVSR_CHECK_ARG(ok16, "ContextBlock fusion with a 16-bit destination needs the lean 3x3 path, no residuals");
    a.gc16 = 1;
    a.dst16 = 0;
  }
  FCVSR_CHECK_ARG(!a.res16 || lean, "16-bit residuals are only supported by the lean 3x3 path");
  {
    const char* e = getenv("FCVSR_MFMA_LEAN");
    if (e && atoi(e) == 0) lean = false;
  }
  // lean 1x1 (flat) path: every source a multiple of 64 channels, channel-contiguous destination, f32 residuals;
  // pixel shuffle only without residuals
  bool lean1 = dst_native && d0.kh == 1 && mw == 1 && !wd && nt <= 64 && (d0.cout % 8 == 0) && !a.planar;
  for (int s2 = 0; s2 < d0.n_src && lean1; ++s2) lean1 = lean1 && (d0.src[s2].c % 64 == 0);
  lean1 = lean1 && (!a.ps || (d0.n_res == 0 && (d0.cout / 4) % 8 == 0));
  for (int g = 0; g < n_groups && lean1; ++g) {
    const fcvsr_conv_desc& d = descs[g];
    long long maxsx = d.dst.sx;
    for (int s2 = 0; s2 < d.n_src; ++s2) maxsx = d.src[s2].sx > maxsx ? d.src[s2].sx : maxsx;
    lean1 = lean1 && d.dst.sc == 1 && (long long)d.B * d.H * d.W * maxsx * 4 < (1ll << 31);
    for (int q = 0; q < d.n_res; ++q)
      lean1 = lean1 && d.res[q].sc == 1 && d.res[q].dtype == FCVSR_F32 && (long long)d.B * d.H * d.W * d.res[q].sx < (1ll << 31);
  }
  {
    const char* e2 = getenv("FCVSR_MFMA_LEAN");
    if (e2 && atoi(e2) == 0) lean1 = false;
  }
  hipError_t e;
  if (lean1) {
    FCVSR_NOTE_KERNEL("conv1_lean_kernel<%s, %d, %s, %s, %s>", tf(mma_dtype == FCVSR_BF16), nt, tf(a.src16), tf(a.dst16), tf(a.ps));
    e = (mma_dtype == FCVSR_BF16) ? dispatch_lean1<true>(a, nt, tiles, st) : dispatch_lean1<false>(a, nt, tiles, st);
    if (e != hipSuccess) {
      set_error("fcvsr_conv2d_mfma: launch failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    return 0;
  }
  // LDS-resident-weight persistent kernel (conv_res.hip): one dense 16-bit source of 64 or 128 channels, cout a multiple of
  // 64, channel-contiguous 16-byte-aligned destination and residuals, no ContextBlock fusion.  It pays once every workgroup
  // amortises its 72 KiB weight copy over a few 8 x 32 tiles; small launches stay on the lean kernel.
  // Pixel-shuffled layers (the 3x3 up-convs of the full / RGB models, 64 -> 256) qualify too: with sub-pixel-major rows a
  // 64-cout block is one sub-pixel, so PixelShuffle is only a different destination pixel (no residuals, 16-bit destination).
  const bool res_ps = a.ps && dst_native && d0.kh == 3 && d0.stride == 1 && mw == 1 && !wd && d0.n_src == 1 && !a.planar && cin == 64 &&
                      d0.cout % 256 == 0 && d0.n_res == 0 && a.dst16 && a.src16 && d0.gc_wmask == nullptr;
  bool res = (lean || res_ps) && a.src16 && conv3_res_supports(cin, d0.cout) && d0.gc_wmask == nullptr && d0.cout == d0.cout / 64 * 64;
  int rtiles = 0;
  for (int g = 0; g < n_groups && res; ++g) {
    const fcvsr_conv_desc& d = descs[g];
    const int dg = a.dst16 ? 8 : 4;
    res = res && d.dst.sc == 1 && d.dst.sx % dg == 0 && d.dst.sy % dg == 0 && d.dst.sb % dg == 0 && ((uintptr_t)d.dst.ptr % 16) == 0;
    res = res && d.src[0].sx % 8 == 0 && d.src[0].sy % 8 == 0 && d.src[0].sb % 8 == 0;
    for (int q = 0; q < d.n_res; ++q) {
      const int rg = d.res[q].dtype == FCVSR_F32 ? 4 : 8;
      res = res && d.res[q].sc == 1 && d.res[q].sx % rg == 0 && d.res[q].sy % rg == 0 && d.res[q].sb % rg == 0 &&
            ((uintptr_t)d.res[q].ptr % 16) == 0;
    }
    rtiles += conv3_res_tiles(d.B, d.H, d.W);
    // (destination element offsets are formed in 32-bit arithmetic and widened before the byte scaling: < 2^30 elements keeps
    // every intermediate positive; 16 clips of 720 x 1280 x 64 are 0.94 * 2^30)
    if (res_ps) res = res && d.dst.c == d0.cout / 4 && (long long)d.B * d.dst.sb < (1ll << 30) && (long long)d.B * d.src[0].sb < (1ll << 29) &&
                      d.src[0].sc == 1 && ((uintptr_t)d.src[0].ptr % 16) == 0;
  }
  {
    // FCVSR_MFMA_RES: 0 never, 1 always (when eligible), unset: by size (FCVSR_MFMA_RES_MIN workgroup-tiles)
    const char* e3 = getenv("FCVSR_MFMA_RES");
    const int res_mode = e3 ? (atoi(e3) ? 1 : 0) : 2;
    const char* e4 = getenv("FCVSR_MFMA_RES_MIN");
    const int res_min = e4 ? atoi(e4) : 768;
    if (res_mode == 0 || (res_mode == 2 && rtiles * (cin == 64 ? d0.cout / 64 : d0.cout / 32) < res_min)) res = false;
  }
  if (res) {
    // 256 zero bytes per DEVICE (the source of halo pixels outside the image), created on first use.  hipMalloc / hipMemset are
    // not capturable: the engine runs every configuration eagerly before it captures a hipGraph, so this never happens inside
    // a capture; a capture that does reach it fails loudly here rather than reading another device's page.
    static void* zeros_dev[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("fcvsr_conv2d_mfma: bad device index"); return FCVSR_E_ARG; }
    if (!zeros_dev[dev]) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        set_error("fcvsr_conv2d_mfma: first use on this device inside a stream capture (run the layer once eagerly first)");
        return FCVSR_E_ARG;
      }
      void* z = nullptr;
      hipError_t ez = hipMalloc(&z, 256);
      if (ez == hipSuccess) ez = hipMemset(z, 0, 256);
      if (ez != hipSuccess) {
        set_error("fcvsr_conv2d_mfma: zero page allocation failed: %s", hipGetErrorString(ez));
        return (int)ez;
      }
      zeros_dev[dev] = z;
    }
    void* zeros = zeros_dev[dev];
    ResArgs wa;
    wa.n_groups = n_groups;
    int wtiles = 0;
    for (int g = 0; g < 3; ++g) {
      const MGroup& G = a.g[g < n_groups ? g : 0];
      ResGroup& Wg = wa.g[g];
      Wg.src = G.src[0]; Wg.res[0] = G.res[0]; Wg.res[1] = G.res[1]; Wg.dst = G.dst; Wg.gc_partial = nullptr;
      Wg.B = G.B; Wg.H = G.H; Wg.W = G.W;
      Wg.tiles_x = cdiv(G.W, 32);
      Wg.tiles_y = cdiv(G.H, 8);
      Wg.tile_begin = wtiles;
      if (g < n_groups) wtiles += G.B * Wg.tiles_x * Wg.tiles_y;
    }
    wa.total_tiles = wtiles;
    wa.cin = cin; wa.cout = d0.cout; wa.cout_pad = d0.cout_pad; wa.cin_pad = a.cin_pad;
    wa.w = a.w; wa.bias = a.bias; wa.act = a.act; wa.slope = a.slope; wa.slope_ptr = a.slope_ptr;
    wa.rs[0] = a.rs[0]; wa.rs[1] = a.rs[1]; wa.n_res = a.n_res; wa.res16 = a.res16; wa.ps = a.ps; wa.zeros = zeros;
    { const char* wd_ = getenv("FCVSR_RES_DBG"); wa.dbg = wd_ ? atoi(wd_) : 0; }
    { const char* wv_ = getenv("FCVSR_RES_V"); wa.variant = wv_ ? atoi(wv_) : 1; }
    wa.stamps = nullptr;
    {
      const char* ws_ = getenv("FCVSR_RES_STAMPS");       // diagnostic: in-kernel cycle stamps of one workgroup (scripts/res_stamps.py)
      if (ws_ && atoi(ws_)) {
        if (!g_res_stamps && hipMalloc(&g_res_stamps, kResStampBytes) != hipSuccess) g_res_stamps = nullptr;
        if (g_res_stamps) (void)hipMemsetAsync(g_res_stamps, 0, kResStampBytes, st);
        wa.stamps = (unsigned long long*)g_res_stamps;
      }
    }
    FCVSR_NOTE_KERNEL((wa.variant >= 3 && mma_dtype == FCVSR_BF16 && a.dst16 && a.n_res == 0) ? "conv3_res3_kernel<%s, %d, %d, %d>" : "conv3_res_kernel<%s, %d, %d, %d>", tf(mma_dtype == FCVSR_BF16), !a.dst16 ? 0 : (a.n_res == 0 ? 2 : 1), cin / 64,
                      a.act == FCVSR_ACT_NONE ? 2 : ((a.act == FCVSR_ACT_RELU || (a.act == FCVSR_ACT_LEAKY && a.slope >= 0.f && a.slope <= 1.f)) ? 1 : 0));
    e = launch_conv3_res(wa, mma_dtype == FCVSR_BF16, a.dst16 != 0, st);
    if (e != hipSuccess) {
      set_error("fcvsr_conv2d_mfma: resident-weight launch failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    return 0;
  }
  if (lean && nt > 64) {      // measured: two 64-cout workgroups per tile beat one 128-cout workgroup (register pressure)
    nt = 64;
    a.n_nblk = (d0.cout + nt - 1) / nt;
  }
  if (lean) FCVSR_NOTE_KERNEL("conv3_lean_kernel<%s, %d, %s, %s>%s", tf(mma_dtype == FCVSR_BF16), nt, tf(a.src16), tf(a.dst16), a.gc_wmask ? " +gc" : "");
  else FCVSR_NOTE_KERNEL("conv_mfma_kernel<%s, %d, %d, %d, %s>", tf(mma_dtype == FCVSR_BF16), nt, d0.kh, mw, tf(wd));
  if (lean) e = (mma_dtype == FCVSR_BF16) ? dispatch_lean<true>(a, nt, tiles, st) : dispatch_lean<false>(a, nt, tiles, st);
  else e = (mma_dtype == FCVSR_BF16) ? dispatch<true>(a, nt, d0.kh, mw, wd, tiles, st) : dispatch<false>(a, nt, d0.kh, mw, wd, tiles, st);
  if (e != hipSuccess) {
    set_error("fcvsr_conv2d_mfma: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
