// Fused IAC step (reference CVSR_freq.py:1230-1250, one loop iteration):
//     out = LeakyReLU_0.1( SAC_h( SAC_v( flow_warp(prev, off), K1 ), K1 ) + feat_in )
// i.e. bilinear warp (:1188-1227), vertical 3-tap adaptive conv, horizontal 3-tap adaptive conv with kernel1 again (:1273),
// residual and activation - one kernel instead of three, the warped tile `s` and the vertical result `v` never leave LDS.
//
// Workgroup = 4 x 16 output pixels x 32 channels.  Phase 1 warps the (4+2) x (16+2) halo tile (replicate padding = clamped
// coordinates) into LDS, phase 2 produces v on 4 x (16+2), phase 3 the output.  Lanes run over (pixel, channel quad): the 8
// quads of a pixel are 8 consecutive lanes, so feature accesses are 128-byte runs and the per-pixel adaptive kernels
// (12 values per quad, channel index c*3+t) are 192/384-byte runs.  HBM-bound: the dominant traffic is K1
// (3*C values per pixel per step), read once from HBM (second touch hits L2) in f32 or the 16-bit MFMA dtype.
#include <stdlib.h>
#include "common.h"
#include "mfma_util.h"

namespace fcvsr {

typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

constexpr int kIY = 4, kIX = 16, kIC = 32;            // tile rows, cols, channels per workgroup
constexpr int kIHX = kIX + 2, kIHY = kIY + 2;

template <int KDT>
__device__ __forceinline__ void load_k12(const void* base, long long elem_off, float k[4][3]) {
  if (KDT == FCVSR_F32) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off);
    const float4 a = p[0], b = p[1], c = p[2];
    k[0][0] = a.x; k[0][1] = a.y; k[0][2] = a.z; k[1][0] = a.w; k[1][1] = b.x; k[1][2] = b.y;
    k[2][0] = b.z; k[2][1] = b.w; k[2][2] = c.x; k[3][0] = c.y; k[3][1] = c.z; k[3][2] = c.w;
  } else {
    const uint2* p = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem_off);
    const uint2 a = p[0], b = p[1], c = p[2];
    const unsigned w[6] = {a.x, a.y, b.x, b.y, c.x, c.y};
    float f[12];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (KDT == FCVSR_BF16) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
      } else {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const h2 hv = __builtin_bit_cast(h2, w[i]);
        f[2 * i] = (float)hv[0];
        f[2 * i + 1] = (float)hv[1];
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < 3; ++t) k[c][t] = f[c * 3 + t];
  }
}

// feature tensors (prev, feat_in, dst) are f32 or 16-bit (ADT); 4 channels per lane
template <int ADT>
__device__ __forceinline__ float4 ld_f4(const float* base, long long elem) {
  if (ADT == FCVSR_F32) return *reinterpret_cast<const float4*>(base + elem);
  const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem);
  if (ADT == FCVSR_BF16)
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 a = __builtin_bit_cast(h2, v.x), b = __builtin_bit_cast(h2, v.y);
  return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}

template <int ADT>
__device__ __forceinline__ void st_f4(float* base, long long elem, float4 x) {
  if (ADT == FCVSR_F32) {
    *reinterpret_cast<float4*>(base + elem) = x;
  } else if (ADT == FCVSR_BF16) {
    typedef __attribute__((ext_vector_type(4))) __bf16 b4;
    const b4 c = {(__bf16)x.x, (__bf16)x.y, (__bf16)x.z, (__bf16)x.w};
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + elem) = __builtin_bit_cast(uint2, c);
  } else {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 c = {(_Float16)x.x, (_Float16)x.y, (_Float16)x.z, (_Float16)x.w};
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + elem) = __builtin_bit_cast(uint2, c);
  }
}

template <int KDT, int ADT>
__global__ __launch_bounds__(256) void iac_step_kernel(View prev, View off, View k1, View fin, float slope, int B, int H,
                                                       int W, View dst, int tiles_x, int tiles_y) {
  __shared__ __align__(16) float s_s[kIHY * kIHX * kIC];
  __shared__ __align__(16) float v_s[kIY * kIHX * kIC];
  const int tid = threadIdx.x;
  const int quad = tid & 7;
  const int c0 = blockIdx.y * kIC + quad * 4;
  const int t = blockIdx.x;
  const int b = t / (tiles_x * tiles_y);
  const int t2 = t - b * tiles_x * tiles_y;
  const int ty0 = (t2 / tiles_x) * kIY, tx0 = (t2 % tiles_x) * kIX;

  // ---- phase 1: s = flow_warp(prev, off) on the halo tile (coordinates clamped = replicate padding of s) -------------
  const long long pp = (long long)b * prev.sb + c0;
  for (int hp = tid >> 3; hp < kIHY * kIHX; hp += 32) {
    const int hy = hp / kIHX, hx = hp - hy * kIHX;
    int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
    gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
    gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
    const float* op = off.p + (long long)b * off.sb + (long long)gy * off.sy + (long long)gx * off.sx;
    const float fx = (float)gx + op[0];
    const float fy = (float)gy + op[off.sc];
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx1 = fx - x0f, wy1 = fy - y0f;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool sane = (fx > -2.f) && (fx < (float)W + 1.f) && (fy > -2.f) && (fy < (float)H + 1.f);
    const int x0 = sane ? (int)x0f : -4, y0 = sane ? (int)y0f : -4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int xi = x0 + dx, yi = y0 + dy;
        if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
          const float w = (dy ? wy1 : wy0) * (dx ? wx1 : wx0);
          const float4 v = ld_f4<ADT>(prev.p, pp + (long long)yi * prev.sy + (long long)xi * prev.sx);
          acc.x = fmaf(v.x, w, acc.x); acc.y = fmaf(v.y, w, acc.y);
          acc.z = fmaf(v.z, w, acc.z); acc.w = fmaf(v.w, w, acc.w);
        }
      }
    }
    *reinterpret_cast<float4*>(s_s + hp * kIC + quad * 4) = acc;
  }
  __syncthreads();

  // ---- phase 2: v[y][hx] = sum_t s[y+t][hx] * K1[y][clamp(hx)][c*3+t] ---------------------------------------------------
  for (int vp = tid >> 3; vp < kIY * kIHX; vp += 32) {
    const int y = vp / kIHX, hx = vp - y * kIHX;
    int gy = ty0 + y, gx = tx0 + hx - 1;
    gy = gy > H - 1 ? H - 1 : gy;
    gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
    float k[4][3];
    load_k12<KDT>(k1.p, (long long)b * k1.sb + (long long)gy * k1.sy + (long long)gx * k1.sx + c0 * 3, k);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
      const float4 v = *reinterpret_cast<const float4*>(s_s + ((y + tt) * kIHX + hx) * kIC + quad * 4);
      acc.x = fmaf(v.x, k[0][tt], acc.x); acc.y = fmaf(v.y, k[1][tt], acc.y);
      acc.z = fmaf(v.z, k[2][tt], acc.z); acc.w = fmaf(v.w, k[3][tt], acc.w);
    }
    *reinterpret_cast<float4*>(v_s + vp * kIC + quad * 4) = acc;
  }
  __syncthreads();

  // ---- phase 3: out = lrelu( sum_t v[y][x+t] * K1[y][x][c*3+t] + feat_in ) ------------------------------------------------
  for (int p = tid >> 3; p < kIY * kIX; p += 32) {
    const int y = p / kIX, x = p - y * kIX;
    const int gy = ty0 + y, gx = tx0 + x;
    if (gy < H && gx < W) {
      float k[4][3];
      load_k12<KDT>(k1.p, (long long)b * k1.sb + (long long)gy * k1.sy + (long long)gx * k1.sx + c0 * 3, k);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        const float4 v = *reinterpret_cast<const float4*>(v_s + (y * kIHX + x + tt) * kIC + quad * 4);
        acc.x = fmaf(v.x, k[0][tt], acc.x); acc.y = fmaf(v.y, k[1][tt], acc.y);
        acc.z = fmaf(v.z, k[2][tt], acc.z); acc.w = fmaf(v.w, k[3][tt], acc.w);
      }
      const float4 f = ld_f4<ADT>(fin.p, (long long)b * fin.sb + (long long)gy * fin.sy + (long long)gx * fin.sx + c0);
      acc.x += f.x; acc.y += f.y; acc.z += f.z; acc.w += f.w;
      acc.x = acc.x >= 0.f ? acc.x : acc.x * slope; acc.y = acc.y >= 0.f ? acc.y : acc.y * slope;
      acc.z = acc.z >= 0.f ? acc.z : acc.z * slope; acc.w = acc.w >= 0.f ? acc.w : acc.w * slope;
      st_f4<ADT>(dst.p, (long long)b * dst.sb + (long long)gy * dst.sy + (long long)gx * dst.sx + c0, acc);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 64-channel variant (C % 64 == 0): a lane owns 8 channels of a pixel, so feature accesses are 16-byte (16-bit storage) or
// 2 x 16-byte (f32) loads in full 128/256-byte runs per pixel and the adaptive kernels are 3 x 16-byte loads in 384-byte runs.
// Every load that does not depend on the offsets (K1 of the lane's pixels, feat_in) is issued before the warp phase, and
// phases 2 and 3 use the same lane <-> pixel mapping so K1 is fetched once and kept (packed) in registers.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kJC = 64;

template <int DT>
struct Pack8 {                                   // 8 channels of a feature tensor as loaded from memory
  uint4 a, b;                                    // f32: a, b = 8 floats; 16-bit: a only
};

template <int DT>
__device__ __forceinline__ Pack8<DT> ld_p8(const float* base, long long elem) {
  Pack8<DT> r;
  if (DT == FCVSR_F32) {
    const uint4* p = reinterpret_cast<const uint4*>(base + elem);
    r.a = p[0]; r.b = p[1];
  } else {
    r.a = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + elem);
    r.b = make_uint4(0, 0, 0, 0);
  }
  return r;
}

template <int DT>
__device__ __forceinline__ void cvt2(unsigned w, float& lo, float& hi) {
  if (DT == FCVSR_BF16) {
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xffff0000u);
  } else {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 hv = __builtin_bit_cast(h2, w);
    lo = (float)hv[0];
    hi = (float)hv[1];
  }
}

template <int DT>
__device__ __forceinline__ void unpack8(const Pack8<DT>& r, float* o) {
  if (DT == FCVSR_F32) {
    o[0] = __uint_as_float(r.a.x); o[1] = __uint_as_float(r.a.y); o[2] = __uint_as_float(r.a.z); o[3] = __uint_as_float(r.a.w);
    o[4] = __uint_as_float(r.b.x); o[5] = __uint_as_float(r.b.y); o[6] = __uint_as_float(r.b.z); o[7] = __uint_as_float(r.b.w);
  } else {
    cvt2<DT>(r.a.x, o[0], o[1]); cvt2<DT>(r.a.y, o[2], o[3]); cvt2<DT>(r.a.z, o[4], o[5]); cvt2<DT>(r.a.w, o[6], o[7]);
  }
}

template <int KDT>
struct PackK {                                   // 24 adaptive-kernel values (8 channels x 3 taps, index c*3+t)
  uint4 q[KDT == FCVSR_F32 ? 6 : 3];
};

template <int KDT>
__device__ __forceinline__ PackK<KDT> ld_k24(const void* base, long long elem) {
  PackK<KDT> r;
  if (KDT == FCVSR_F32) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(base) + elem);
#pragma unroll
    for (int i = 0; i < 6; ++i) r.q[i] = p[i];
  } else {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + elem);
#pragma unroll
    for (int i = 0; i < 3; ++i) r.q[i] = p[i];
  }
  return r;
}

template <int KDT>
__device__ __forceinline__ void unpack_k24(const PackK<KDT>& r, float* k) {
  if (KDT == FCVSR_F32) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      k[4 * i] = __uint_as_float(r.q[i].x); k[4 * i + 1] = __uint_as_float(r.q[i].y);
      k[4 * i + 2] = __uint_as_float(r.q[i].z); k[4 * i + 3] = __uint_as_float(r.q[i].w);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      cvt2<KDT>(r.q[i].x, k[8 * i], k[8 * i + 1]); cvt2<KDT>(r.q[i].y, k[8 * i + 2], k[8 * i + 3]);
      cvt2<KDT>(r.q[i].z, k[8 * i + 4], k[8 * i + 5]); cvt2<KDT>(r.q[i].w, k[8 * i + 6], k[8 * i + 7]);
    }
  }
}

template <int ADT>
__device__ __forceinline__ void st_p8(float* base, long long elem, const float* x) {
  if (ADT == FCVSR_F32) {
    *reinterpret_cast<float4*>(base + elem) = make_float4(x[0], x[1], x[2], x[3]);
    *reinterpret_cast<float4*>(base + elem + 4) = make_float4(x[4], x[5], x[6], x[7]);
  } else if (ADT == FCVSR_BF16) {
    typedef __attribute__((ext_vector_type(8))) __bf16 b8;
    const b8 c = {(__bf16)x[0], (__bf16)x[1], (__bf16)x[2], (__bf16)x[3], (__bf16)x[4], (__bf16)x[5], (__bf16)x[6], (__bf16)x[7]};
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + elem) = __builtin_bit_cast(uint4, c);
  } else {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    const h8 c = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3], (_Float16)x[4], (_Float16)x[5], (_Float16)x[6], (_Float16)x[7]};
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + elem) = __builtin_bit_cast(uint4, c);
  }
}

struct IacDir {
  View prev, off, fin, dst;
};
struct IacArgs {
  IacDir d[2];            // forward / backward alignment direction (ND = 1 uses d[0] only)
  View k1;                // FK = false: the 3*C adaptive-kernel channels of this iteration
  View k0;                // FK = true: the 64-channel input of the last kernel-predictor layer (F[1], a 1x1 convolution) ...
  const uint16_t* wk;     // ... its 192 x 64 weight rows of this iteration (MFMA dtype, cin contiguous) ...
  const float* kbias;     // ... and their bias
  float slope;
  int B, H, W, tiles_x, tiles_y;
  int ntiles;             // B * tiles_x * tiles_y: the persistent workgroups of iac_step64_kernel split this range
};

constexpr int kKtRow = 3 * kJC + 8;                   // halfwords per pixel row of the predicted-kernel tile in LDS

// ND = 2: both alignment directions of one IAC iteration in one launch.  They use the SAME adaptive kernels
// (reference :1524-1545 passes Pred_K to IAC for the forward and the backward group alike), so K1 - half of the bytes
// a step reads - is fetched once and serves both.
// FK = true: the adaptive kernels are never materialised in HBM.  The last layer of the kernel predictor is a 1x1
// convolution (reference :1416, F[1]), so the 3*C kernel values of the tile's 4 x 18 pixels are one small GEMM
// (192 x 64) x (64 x 72): 18 MFMA tiles spread over the 4 waves, operands straight from L2 (weights) / HBM (the 64-channel
// predictor features), result rounded to the MFMA dtype into LDS - exactly what the stand-alone F[1] launch would have
// stored, minus 1152 bytes written and 2 x 1152 bytes read per pixel and iteration set.
#ifndef FCVSR_IAC_PIPE
#define FCVSR_IAC_PIPE 1
#endif
// Ablation builds (-DFCVSR_IAC_ABL=bits, one library per value run through scripts/ab_lib.sh; profiles/r03_iac_ablation.txt):
// 1 no gather loads, 2 no SAC arithmetic / LDS reads, 4 no predictor GEMM, 8 no stores, 16 no offset loads.  0 in every shipped build.
#ifndef FCVSR_IAC_ABL
#define FCVSR_IAC_ABL 0
#endif
// Persistent workgroups (the host launches two per CU): each walks a contiguous run of tiles, so the predictor weights of
// its waves (FK: 4 or 8 MFMA A-fragments) and the bias are fetched once per workgroup instead of once per tile (24 KB per
// tile was a quarter of what a tile pulled through L2), and consecutive tiles of a workgroup share their halo columns in L1.
template <int KDT, int ADT, int ND, bool FK>
__global__ __launch_bounds__(256, 2) void iac_step64_kernel(IacArgs a) {
  constexpr bool PIPE = FCVSR_IAC_PIPE != 0;
  const View k1 = a.k1;
  const float slope = a.slope;
  const int H = a.H, W = a.W, tiles_x = a.tiles_x, tiles_y = a.tiles_y;
  // LDS: the warped tile s (6 x 18 pixels) and the vertical result v (4 x 18), f32, 64 channels each (46,080 bytes).  The
  // predicted-kernel tile (FK, 28,800 bytes) lives in s's bytes (+1,152 of its own): it is written by the GEMM, copied into
  // the owning lanes' registers and dead before the first warped value is stored (two barriers, see below).  FK adds the
  // predictor's 192 x 64 weights (rows padded to 144 bytes: the 16 rows of a ds_read_b128 lane group fall on 16 different
  // slots) and bias, resident for the workgroup's whole tile run: 75,648 bytes, two workgroups per CU.
  constexpr int kSBytes = kIHY * kIHX * kJC * 4, kVBytes = kIY * kIHX * kJC * 4, kKtBytes = kIY * kIHX * kKtRow * 2;
  constexpr int kWRow = kJC + 8, kWBytes = FK ? 3 * kJC * kWRow * 2 : 0, kKbBytes = FK ? 3 * kJC * 4 : 0;
  constexpr int kS2Bytes = FK && kKtBytes > kSBytes ? kKtBytes : kSBytes;      // s_s and kt_s share these bytes, v_s does not
  __shared__ __align__(16) unsigned char smem[kS2Bytes + kVBytes + kWBytes + kKbBytes];
  float* const s_s = reinterpret_cast<float*>(smem);
  float* const v_s = reinterpret_cast<float*>(smem + kS2Bytes);
  uint16_t* const kt_s = reinterpret_cast<uint16_t*>(smem);
  uint16_t* const w_s = reinterpret_cast<uint16_t*>(smem + kS2Bytes + kVBytes);
  float* const kb_s = reinterpret_cast<float*>(smem + kS2Bytes + kVBytes + kWBytes);
  // s / v pixel records are 64 floats = one 256-byte bank row.  A lane's 8 channels are kept as two 16-byte halves 128
  // bytes apart ([half][oct][4]) with the halves swapped on odd pixels: the 8 lanes of a pixel then write 128 contiguous
  // bytes per store, and the four pixels x four lane-quads of a ds_read_b128 group land on 16 different slots (the plain
  // [oct][8] record put lanes l and l+4 of every store and two of the four pixels of every read group on the same banks).
  constexpr bool BF = KDT == FCVSR_BF16;
  int t_begin, t_end;
  {                                                    // workgroups go round-robin to the 8 XCDs: give each XCD a contiguous
    const int g = blockIdx.x;                          // run of tiles so halos hit its L2
    const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = g & 7, loc = g >> 3;
    const int ci = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    t_begin = (int)((long long)ci * a.ntiles / nwg);
    t_end = (int)((long long)(ci + 1) * a.ntiles / nwg);
  }
  if (FK) {                                            // predictor weights and bias: once per workgroup
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 3 * kJC * kJC / 8 / 256; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, seg = idx & 7;
      *reinterpret_cast<uint4*>(w_s + row * kWRow + seg * 8) = *reinterpret_cast<const uint4*>(a.wk + row * kJC + seg * 8);
    }
    if (tid < 3 * kJC) kb_s[tid] = a.kbias[tid];
    __syncthreads();
  }
  // The offsets head the longest dependency chain of a tile (offsets -> tap addresses -> gather -> warp).  A persistent
  // workgroup also pays for its own stores: vmcnt retires in order, so the first wait of a tile would sit behind the previous
  // tile's output stores.  The next tile's offsets are therefore requested before those stores are issued.
  auto load_off = [&](const int tt, const int tidv) {
    const int lanev = tidv & 63, wavev = tidv >> 6;
    const int bb = tt / (tiles_x * tiles_y), tt2 = tt - bb * tiles_x * tiles_y;
    const int oy0 = (tt2 / tiles_x) * kIY, ox0 = (tt2 % tiles_x) * kIX;
    const int sdir = (lanev >> 5) < ND ? (lanev >> 5) : 0, sit = (lanev >> 3) & 3, spl = lanev & 7;
    int hp = sit * 32 + wavev * 8 + spl;
    hp = hp < kIHY * kIHX ? hp : kIHY * kIHX - 1;
    const int hy = hp / kIHX, hx = hp - hy * kIHX;
    int gy = oy0 + hy - 1, gx = ox0 + hx - 1;
    gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
    gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
    // a.d[sdir] with a lane-dependent sdir is a VECTOR load from the kernel-argument block (and a round trip of its own
    // before the offsets can even be requested): select between the two directions' scalar fields instead
    const bool d1 = sdir != 0;
    const float* op0 = d1 ? a.d[1].off.p : a.d[0].off.p;
    const long long osb = d1 ? a.d[1].off.sb : a.d[0].off.sb, osy = d1 ? a.d[1].off.sy : a.d[0].off.sy;
    const long long osx = d1 ? a.d[1].off.sx : a.d[0].off.sx, osc = d1 ? a.d[1].off.sc : a.d[0].off.sc;
    const float* op = op0 + (long long)bb * osb + (long long)gy * osy + (long long)gx * osx;
    if (FCVSR_IAC_ABL & 16) return make_float2(0.25f, 0.25f);
    return make_float2(op[0], op[osc]);
  };
  float2 off_next = make_float2(0.f, 0.f);
  if (t_begin < t_end) off_next = load_off(t_begin, threadIdx.x);
  for (int t = t_begin; t < t_end; ++t) {
  // Everything derived from the lane id is recomputed per tile (a handful of VALU ops): hoisted out of the loop these
  // values - LDS record offsets, channel offsets, fragment rows - cost two dozen registers the tile body has no room for
  // (hipcc spilled them to scratch).  The empty asm makes the id opaque, so nothing below is loop-invariant.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int oct = tid & 7, ps = tid >> 3;              // 8-channel group, pixel slot (32 per pass)
  auto rec = [&](const int pix, const int h) { return pix * kJC + ((h ^ (pix & 1)) << 5) + oct * 4; };
  const int c0 = blockIdx.y * kJC + oct * 8;
  const int lane = tid & 63, wave = tid >> 6, mr = lane & 31, hh = lane >> 5;
  const int b = t / (tiles_x * tiles_y);
  const int t2 = t - b * tiles_x * tiles_y;
  const int ty0 = (t2 / tiles_x) * kIY, tx0 = (t2 % tiles_x) * kIX;

  // ---- loads that do not depend on the offsets: K1 and feat_in of the lane's two interior pixels, K1 of its halo column ----
  // interior pixel of pass j: p = j*32 + ps -> (y, x) = (p >> 4, p & 15); halo-column pixel (ps < 8): y = ps >> 1, hx = 0 or 17
  PackK<KDT> kin[2], khal;
  if (!FK) {
    const long long kb = (long long)b * k1.sb + c0 * 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = j * 32 + ps;
      int gy = ty0 + (p >> 4), gx = tx0 + (p & 15);
      gy = gy > H - 1 ? H - 1 : gy;
      gx = gx > W - 1 ? W - 1 : gx;
      kin[j] = ld_k24<KDT>(k1.p, kb + (long long)gy * k1.sy + (long long)gx * k1.sx);
    }
    {
      int gy = ty0 + (ps >> 1), gx = tx0 + ((ps & 1) ? kIX : -1);
      gy = gy > H - 1 ? H - 1 : gy;
      gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
      if (ps < 8) khal = ld_k24<KDT>(k1.p, kb + (long long)gy * k1.sy + (long long)gx * k1.sx);
    }
  }
  // ---- kernel-predictor GEMM (FK): D[cout][pixel] = W[cout][:] . k0[pixel][:] + bias, 6 cout tiles x 3 pixel tiles.
  // The 12 pixel fragments are requested here, with the offsets; the MFMAs run after the first direction's taps have been
  // issued, so the gather latency hides behind them.
  uint4 kf[3][4];
  if (FK && !(FCVSR_IAC_ABL & 4)) {
    const uint16_t* k0p = reinterpret_cast<const uint16_t*>(a.k0.p) + (long long)b * a.k0.sb + hh * 8;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      int p = nt * 32 + mr;
      p = p < kIY * kIHX ? p : kIY * kIHX - 1;
      const int y = p / kIHX, hx = p - y * kIHX;
      int gy = ty0 + y, gx = tx0 + hx - 1;
      gy = gy > H - 1 ? H - 1 : gy;
      gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
      const uint16_t* kp = k0p + (long long)gy * a.k0.sy + (long long)gx * a.k0.sx;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[nt][kk] = *reinterpret_cast<const uint4*>(kp + kk * 16);
    }
  }
  auto predictor_gemm = [&]() {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int mt = wave + 4 * q;
      if (mt < 6) {                                    // wave-uniform: wave w owns cout tile w, waves 0 and 1 also 4 and 5
        uint4 wfr[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wfr[kk] = *reinterpret_cast<const uint4*>(w_s + (mt * 32 + mr) * kWRow + hh * 8 + kk * 16);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
          f32x16_t acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc = mfma<BF>(wfr[kk], kf[nt][kk], acc);
          const int p = nt * 32 + mr;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co = mt * 32 + 8 * g + 4 * hh;   // acc[4g..4g+3] = couts co..co+3 of pixel column mr
            const float4 b4 = *reinterpret_cast<const float4*>(kb_s + co);
            const uint2 pk = cvt4<BF>(make_float4(acc[4 * g] + b4.x, acc[4 * g + 1] + b4.y, acc[4 * g + 2] + b4.z, acc[4 * g + 3] + b4.w));
            if (p < kIY * kIHX) *reinterpret_cast<uint2*>(kt_s + p * kKtRow + co) = pk;
          }
        }
      }
    }
  };

  // ---- software pipeline over the (up to two) directions: every load of direction 1 that does not depend on LDS is in
  // flight while direction 0 runs its LDS phases -------------------------------------------------------------------------
  constexpr int NP = (kIHY * kIHX + 31) / 32;
  static_assert(NP * 8 * ND <= 64, "one sampling slot per lane");
  // ---- sampling set-up, one (direction, iteration, pixel) slot per LANE.  A wave covers 8 halo pixels per iteration and
  // the 8 lanes of a pixel need the same offsets / tap addresses / tap weights: instead of every lane redoing the ~75
  // instructions for each of its NP x ND pixels, lane l computes slot (dir, it, pixel) = (l >> 5, (l >> 3) & 3, l & 7) once
  // and the values are handed out with wave shuffles below (no LDS round trip, no barrier).
  const int wv = wave;
  int my_eo[4];
  float my_w[4];
  {
    const int sdir = (lane >> 5) < ND ? (lane >> 5) : 0, sit = (lane >> 3) & 3, spl = lane & 7;
    int hp = sit * 32 + wv * 8 + spl;
    hp = hp < kIHY * kIHX ? hp : kIHY * kIHX - 1;
    const int hy = hp / kIHX, hx = hp - hy * kIHX;
    int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
    gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
    gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
    const int psy = (int)(sdir ? a.d[1].prev.sy : a.d[0].prev.sy);    // scalar fields selected per lane (no kernarg vector load);
    const int psx = (int)(sdir ? a.d[1].prev.sx : a.d[0].prev.sx);    // 32-bit offsets inside one image (host checks H*sy < 2^31)
    const float fx = (float)gx + off_next.x;
    const float fy = (float)gy + off_next.y;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx1 = fx - x0f, wy1 = fy - y0f;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool sane = (fx > -2.f) && (fx < (float)W + 1.f) && (fy > -2.f) && (fy < (float)H + 1.f);
    const int x0 = sane ? (int)x0f : -4, y0 = sane ? (int)y0f : -4;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int xi = x0 + dx, yi = y0 + dy;
        const bool in = xi >= 0 && xi < W && yi >= 0 && yi < H;
        const int xc = xi < 0 ? 0 : (xi > W - 1 ? W - 1 : xi), yc = yi < 0 ? 0 : (yi > H - 1 ? H - 1 : yi);
        my_w[dy * 2 + dx] = in ? (dy ? wy1 : wy0) * (dx ? wx1 : wx0) : 0.f;   // 0 outside the image = zero padding
        my_eo[dy * 2 + dx] = yc * psy + xc * psx;
      }
    }
  }
  Pack8<ADT> fpk[ND][2];                               // feat_in of the lane's two interior pixels (consumed in phase 3)
  auto issue_fin = [&](const int dir) {
    const View fin = a.d[dir].fin;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = j * 32 + ps;
      int gy = ty0 + (p >> 4), gx = tx0 + (p & 15);
      gy = gy > H - 1 ? H - 1 : gy;
      gx = gx > W - 1 ? W - 1 : gx;
      fpk[dir][j] = ld_p8<ADT>(fin.p, (long long)b * fin.sb + (long long)gy * fin.sy + (long long)gx * fin.sx + c0);
    }
  };
  issue_fin(0);

  // phase 1a: the 4 x NP bilinear taps of a direction (unconditional: out-of-image taps read a clamped address with
  // weight 0); addresses and weights come from the slot owner's registers
  Pack8<ADT> tap[ND][NP][4];
  auto issue_taps = [&](const int dir) {
    const View prev = a.d[dir].prev;
    const long long pp = (long long)b * prev.sb + c0;
    const float* pb = ADT == FCVSR_F32 ? prev.p + pp
                                       : reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(prev.p) + pp);
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int src = dir * 32 + it * 8 + (lane >> 3);  // the lane that owns (dir, it, this lane's pixel)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (FCVSR_IAC_ABL & 1) tap[dir][it][q] = Pack8<ADT>{};
        else tap[dir][it][q] = ld_p8<ADT>(pb, __shfl(my_eo[q], src));
      }
    }
  };
  // phase 1b: s = flow_warp(prev, off) on the halo tile (coordinates clamped = replicate padding of s)
  auto warp_store = [&](const int dir) {
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int hp = it * 32 + ps;
      const int src = dir * 32 + it * 8 + (lane >> 3);
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float tw[4];                                      // fetched here, not with the addresses: 16 registers per direction
#pragma unroll
      for (int q = 0; q < 4; ++q) tw[q] = __shfl(my_w[q], src);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[8];
        unpack8<ADT>(tap[dir][it][q], v);
        // out-of-image taps carry weight 0 (their clamped sample is a finite in-image value)
        const float w = tw[q];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = fmaf(v[c], w, acc[c]);
      }
      if (hp < kIHY * kIHX) {
        *reinterpret_cast<float4*>(s_s + rec(hp, 0)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(s_s + rec(hp, 1)) = make_float4(acc[4], acc[5], acc[6], acc[7]);
      }
    }
  };
  // phases 2 and 3 of a direction (s_s complete on entry)
  auto sac_phases = [&](const int dir) {
    const View dst = a.d[dir].dst;
    // ---- phase 2: v[y][hx] = sum_t s[y+t][hx] * K1[y][clamp(hx)][c*3+t] -------------------------------------------------
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int y, hx;
      if (j < 2) { const int p = j * 32 + ps; y = p >> 4; hx = (p & 15) + 1; }
      else { y = ps >> 1; hx = (ps & 1) ? kIHX - 1 : 0; }
      if (j < 2 || ps < 8) {
        float k[24];
        unpack_k24<KDT>(j < 2 ? kin[j] : khal, k);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < ((FCVSR_IAC_ABL & 2) ? 0 : 3); ++tt) {
          const int sp = (y + tt) * kIHX + hx;
          const float4 va = *reinterpret_cast<const float4*>(s_s + rec(sp, 0)), vb = *reinterpret_cast<const float4*>(s_s + rec(sp, 1));
          const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] = fmaf(v[c], k[c * 3 + tt], acc[c]);
        }
        const int vp = y * kIHX + hx;
        *reinterpret_cast<float4*>(v_s + rec(vp, 0)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(v_s + rec(vp, 1)) = make_float4(acc[4], acc[5], acc[6], acc[7]);
      }
    }
    __syncthreads();
    // ---- phase 3: out = lrelu( sum_t v[y][x+t] * K1[y][x][c*3+t] + feat_in ) ----------------------------------------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = j * 32 + ps;
      const int y = p >> 4, x = p & 15;
      const int gy = ty0 + y, gx = tx0 + x;
      if (gy < H && gx < W) {
        float k[24], f[8];
        unpack_k24<KDT>(kin[j], k);
        unpack8<ADT>(fpk[dir][j], f);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < ((FCVSR_IAC_ABL & 2) ? 0 : 3); ++tt) {
          const int vp = y * kIHX + x + tt;
          const float4 va = *reinterpret_cast<const float4*>(v_s + rec(vp, 0)), vb = *reinterpret_cast<const float4*>(v_s + rec(vp, 1));
          const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] = fmaf(v[c], k[c * 3 + tt], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          acc[c] += f[c];
          acc[c] = acc[c] >= 0.f ? acc[c] : acc[c] * slope;
        }
        if (!(FCVSR_IAC_ABL & 8) || acc[0] == 1234.5f)
          st_p8<ADT>(dst.p, (long long)b * dst.sb + (long long)gy * dst.sy + (long long)gx * dst.sx + c0, acc);
      }
    }
  };

  issue_taps(0);
  if (FK && !(FCVSR_IAC_ABL & 4)) {                                            // the predicted kernels of the lane's pixels: LDS -> registers, once
    predictor_gemm();                                  // kf requested before the taps: in flight since the tile began
    __syncthreads();                                   // GEMM results of all four waves are in kt_s
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = j * 32 + ps;
      const uint4* kq = reinterpret_cast<const uint4*>(kt_s + ((p >> 4) * kIHX + (p & 15) + 1) * kKtRow + oct * 24);
      kin[j].q[0] = kq[0]; kin[j].q[1] = kq[1]; kin[j].q[2] = kq[2];
    }
    if (ps < 8) {
      const uint4* kq = reinterpret_cast<const uint4*>(kt_s + ((ps >> 1) * kIHX + ((ps & 1) ? kIHX - 1 : 0)) * kKtRow + oct * 24);
      khal.q[0] = kq[0]; khal.q[1] = kq[1]; khal.q[2] = kq[2];
    }
    __syncthreads();                                   // kt_s is dead: its bytes become s_s / v_s
  }
  warp_store(0);
  if (ND > 1 && PIPE) { issue_taps(1); issue_fin(1); } // in flight during direction 0's LDS phases
  if (t + 1 < t_end) off_next = load_off(t + 1, tid);
  __syncthreads();
  sac_phases(0);
  if (ND > 1) {
    // s_s was last read in phase 2 of direction 0, which every wave has left (barrier inside sac_phases)
    if (!PIPE) { issue_taps(1); issue_fin(1); }
    warp_store(1);
    __syncthreads();
    sac_phases(1);
  }
  // Next tile: s_s is rewritten (by kt_s in the FK case) after its last read (phase 2, before the barrier inside sac_phases);
  // v_s is read until the end of phase 3 and rewritten in the next tile's phase 2, behind the barrier after its warp_store.
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Second form of the fused-predictor step (round 3; 16-bit activations, both directions): the predicted kernels never touch
// LDS.  Lane (n, q) = (l & 15, l >> 4) of wave w owns pixel (row w, column n) of the tile's 4 x 16 "v domain" (14 interior
// columns + one halo column each side) and 16 consecutive channels q*16 .. q*16+15.  The predictor GEMM runs per wave as
// 12 M-tiles of v_mfma_f32_16x16x32 (A = weight rows, B = the row's 16 pixels of k0): with the weight rows permuted so that
// M-tile (tap t, block cb) row r is channel (r >> 2) * 16 + cb * 4 + (r & 3), the accumulator of lane (n, q) IS the kernel of
// its pixel for channels q*16 + cb*4 .. +3 and tap t.  No kernel tile in LDS, no transposing barriers, 8 instead of 48
// registers of predictor input, 24 instead of 36 registers of kernels; v overlays s (the vertical results wait in registers
// across one barrier): 49.9 KB of LDS and < 168 VGPRs = three workgroups per CU.  s / v records are 64 floats in 16 slots of
// 16 bytes, slot index XOR (pixel & 15): the warp's stores (8 lanes x 2 slots per pixel) and the SAC reads (16 pixels x 4
// slots per instruction) both spread evenly over the banks.
constexpr int kQX = 14, kQVX = 16, kQY = 4, kQSY = kQY + 2;
constexpr int kQNS = kQSY * kQVX, kQNV = kQY * kQVX;     // 96 warped pixels, 64 vertical results per tile

template <bool BF16>
__device__ __forceinline__ f32x4v_t mfma16x(uint4 a, uint4 b, f32x4v_t c) {
  if (BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

struct Iac2Args {        // compact: both directions share every stride (host-checked), 32-bit offsets inside one image
  const uint16_t* prev[2];
  const float* off[2];
  const uint16_t* fin[2];
  uint16_t* dst[2];
  const uint16_t* k0;
  const uint16_t* wk;
  const float* kbias;
  long long prev_sb, off_sb, fin_sb, dst_sb, k0_sb;
  int prev_sy, prev_sx, off_sy, off_sx, off_sc, fin_sy, fin_sx, dst_sy, dst_sx, k0_sy, k0_sx;
  float slope;
  int H, W, tiles_x, tiles_y, ntiles;
};

#ifndef FCVSR_IAC2_WGS
#define FCVSR_IAC2_WGS 3
#endif
#ifndef FCVSR_IAC2_EARLY
#define FCVSR_IAC2_EARLY 2
#endif
// (Tried and dropped, round 3: 8-wave workgroups - two 4-row tiles over one weight copy, 16 waves per CU: 252 vs 248 us; one 8-row tile
// (warped halo 1.43 instead of 1.71 pixels per output pixel) needs <= 128 registers for two workgroups per CU and spilled 49.)
template <int KDT, int ADT>
__global__ __launch_bounds__(256, FCVSR_IAC2_WGS) void iac_fused2_kernel(Iac2Args a) {
  constexpr int NG = 1;
  static_assert(ADT != FCVSR_F32 && KDT != FCVSR_F32, "16-bit form");
  constexpr bool BF = KDT == FCVSR_BF16;
  constexpr int SPP = 32 * NG;                           // pixel slots per gather pass
  constexpr int ND = 2, NP = (kQNS + SPP - 1) / SPP;     // 3 gather passes
  constexpr int EARLY = FCVSR_IAC2_EARLY;                // passes of direction 1 requested before direction 0's LDS phases
  __shared__ __align__(16) float s_all[kQNS * kJC];      // 24,576 B (4 rows) / 40,960 B (8 rows); v = the first kQY * 16 records
  __shared__ __align__(16) uint16_t w_s[3 * kJC * kJC];  // 24,576 B, 16-byte segments XOR (row & 7)
  __shared__ __align__(16) float kb_s[3 * kJC];
  const float slope = a.slope;
  const int H = a.H, W = a.W, tiles_x = a.tiles_x, tiles_y = a.tiles_y;
  int t_begin, t_end;
  {
    const int g = blockIdx.x, nwg = gridDim.x, qq = nwg >> 3, rr = nwg & 7, xcd = g & 7, loc = g >> 3;
    const int ci = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
    t_begin = (int)((long long)ci * a.ntiles / nwg);
    t_end = (int)((long long)(ci + 1) * a.ntiles / nwg);
  }
  {                                                      // predictor weights (rows permuted) and bias: once per workgroup
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 3 * kJC * kJC / 8 / (256 * NG); ++i) {
      const int idx = tid + 256 * NG * i, R = idx >> 3, seg = idx & 7;
      const int mt = R >> 4, r = R & 15, tp = mt >> 2, cb = mt & 3;
      const int orig = ((r >> 2) * 16 + cb * 4 + (r & 3)) * 3 + tp;
      *reinterpret_cast<uint4*>(w_s + R * kJC + ((seg ^ (R & 7)) << 3)) = *reinterpret_cast<const uint4*>(a.wk + orig * kJC + seg * 8);
    }
    if (tid < 3 * kJC) {
      const int mt = tid >> 4, r = tid & 15, tp = mt >> 2, cb = mt & 3;
      kb_s[tid] = a.kbias[((r >> 2) * 16 + cb * 4 + (r & 3)) * 3 + tp];
    }
    __syncthreads();
  }
  auto load_off = [&](const int tt, const int tidv) {
    const int lanev = tidv & 63, wavev = tidv >> 6;
    const int bb = tt / (tiles_x * tiles_y), tt2 = tt - bb * tiles_x * tiles_y;
    const int oy0 = (tt2 / tiles_x) * kQY, ox0 = (tt2 % tiles_x) * kQX;
    const int sdir = lanev >> 5, sit = (lanev >> 3) & 3, spl = lanev & 7;
    int hp = sit * SPP + wavev * 8 + spl;
    hp = hp < kQNS ? hp : kQNS - 1;
    const int hy = hp >> 4, hx = hp & 15;
    int gy = oy0 + hy - 1, gx = ox0 + hx - 1;
    gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
    gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
    const float* op = (sdir ? a.off[1] : a.off[0]) + (long long)bb * a.off_sb + (gy * a.off_sy + gx * a.off_sx);
    return make_float2(op[0], op[a.off_sc]);
  };
  const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // row of the tile (scalar)
  float* const s_s = s_all;
  float* const v_s = s_s;
  float2 off_next = make_float2(0.f, 0.f);
  if (t_begin < t_end) off_next = load_off(t_begin, (int)threadIdx.x);
  for (int t = t_begin; t < t_end; ++t) {
    // thread id = scalar wave index * 64 + lane (mbcnt): recomputed per tile, no vector register carried around the loop
    int tid;                                             // volatile asm: not hoisted out of the tile loop (where it would be spilled)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshl_add_u32 %0, %1, 6, %0" : "=&v"(tid) : "s"(wave_s));
    // Every phase derives what it needs from its OWN opaque copy of the thread id (a few VALU ops): one shared set of lane
    // values (pixel coordinates, record addresses, channel offsets) lived across the whole tile and was spilled to scratch -
    // and a scratch reload between the gather requests waits for all of them (vmcnt retires in order).
    auto fresh = [&]() { int x = tid; asm volatile("" : "+v"(x)); return x; };
    const int wave = wave_s;
    const int b = t / (tiles_x * tiles_y);
    const int t2 = t - b * tiles_x * tiles_y;
    const int ty0 = (t2 / tiles_x) * kQY, tx0 = (t2 % tiles_x) * kQX;
    const int row = ty0 + wave;                          // the wave's row of the v domain (scalar)
    const int rowc = row > H - 1 ? H - 1 : row;          // clamped = replicate padding of K1 / feat_in
    // SAC / GEMM role: lane (n, q) = (l & 15, l >> 4) owns pixel (row, tx0 + n - 1), channels q*16 .. q*16+15
    auto col_of = [&](const int n_) { const int x = tx0 + n_ - 1; return x < 0 ? 0 : (x > W - 1 ? W - 1 : x); };
    // uniform base + unsigned 32-bit byte offset: the scalar-base addressing form (one offset register per access, no 64-bit pairs)
    auto ld16 = [](const void* base, const unsigned byte_off) {
      Pack8<ADT> r;
      r.a = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + byte_off);
      r.b = make_uint4(0, 0, 0, 0);
      return r;
    };
    // ---- predictor input of the pixel: cin q*8 .. +7 and 32 + q*8 .. +7 ----
    uint4 kfr[2];
    {
      const int x = fresh(), n = x & 15, q = (x & 63) >> 4;
      const uint16_t* kb = a.k0 + (long long)b * a.k0_sb + (long long)rowc * a.k0_sy;
      const unsigned e = (unsigned)(col_of(n) * a.k0_sx + q * 8) * 2u;
      kfr[0] = ld16(kb, e).a;
      kfr[1] = ld16(kb, e + 64u).a;
    }
    // ---- sampling set-up: lane slot (dir, pass, pixel) = (l >> 5, (l >> 3) & 3, l & 7) of the wave's 8 pixels per pass ----
    int my_eo[4];
    float my_w[4];
    {
      const int x = fresh(), lane = x & 63;
      const int sit = (lane >> 3) & 3, spl = lane & 7;
      int hp = sit * SPP + wave * 8 + spl;
      hp = hp < kQNS ? hp : kQNS - 1;
      const int hy = hp >> 4, hx = hp & 15;
      int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
      gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
      gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
      const int psy = a.prev_sy, psx = a.prev_sx;
      const float fx = (float)gx + off_next.x;
      const float fy = (float)gy + off_next.y;
      const float x0f = floorf(fx), y0f = floorf(fy);
      const float wx1 = fx - x0f, wy1 = fy - y0f;
      const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
      int wb, hb;                                        // moved and converted here, per tile: hoisted out of the loop the two floats were spilled
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(wb), "=&v"(hb) : "s"(W + 1), "s"(H + 1));
      const bool sane = (fx > -2.f) && (fx < (float)wb) && (fy > -2.f) && (fy < (float)hb);
      const int x0 = sane ? (int)x0f : -4, y0 = sane ? (int)y0f : -4;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int xi = x0 + dx, yi = y0 + dy;
          const bool in = xi >= 0 && xi < W && yi >= 0 && yi < H;
          const int xc = xi < 0 ? 0 : (xi > W - 1 ? W - 1 : xi), yc = yi < 0 ? 0 : (yi > H - 1 ? H - 1 : yi);
          my_w[dy * 2 + dx] = in ? (dy ? wy1 : wy0) * (dx ? wx1 : wx0) : 0.f;
          my_eo[dy * 2 + dx] = yc * psy + xc * psx;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);                   // phase fences for the scheduler: hoisting across them cost 30-50 spilled registers
    Pack8<ADT> fpk[2];                                   // feat_in of the lane's pixel, channels q*16 .. +15 (one direction at a time)
    auto issue_fin = [&](const int dir) {
      const int x = fresh(), n = x & 15, q = (x & 63) >> 4;
      const uint16_t* fb = a.fin[dir] + (long long)b * a.fin_sb + (long long)rowc * a.fin_sy;
      const unsigned e = (unsigned)(col_of(n) * a.fin_sx + q * 16) * 2u;
      fpk[0] = ld16(fb, e);
      fpk[1] = ld16(fb, e + 16u);
    };
    Pack8<ADT> tap[ND][NP][4];
    auto issue_taps = [&](const int dir, const int it0, const int it1) {
      const int x = fresh(), oct = x & 7, pull = ((x & 63) >> 3) << 2;   // ds_bpermute byte index of lane (l >> 3)
      const uint16_t* pb = a.prev[dir] + (long long)b * a.prev_sb;
#pragma unroll
      for (int it = it0; it < it1; ++it) {
        const int src = pull + dir * 128 + it * 32;       // + 128 * dir + 32 * pass = the slot's owner
#pragma unroll
        for (int k = 0; k < 4; ++k)
          tap[dir][it][k] = ld16(pb, (unsigned)(__builtin_amdgcn_ds_bpermute(src, my_eo[k]) + oct * 8) * 2u);
      }
    };
    auto warp_store = [&](const int dir) {
      const int x = fresh(), oct = x & 7, ps = x >> 3, pull = ((x & 63) >> 3) << 2;
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int hp = it * SPP + ps;
        const int src = pull + dir * 128 + it * 32;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float tw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) tw[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(my_w[k])));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v[8];
          unpack8<ADT>(tap[dir][it][k], v);
          const float w = tw[k];
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] = fmaf(v[c], w, acc[c]);
        }
        float* rec = s_s + hp * kJC;
        if (kQNS % SPP == 0 || hp < kQNS) {
          *reinterpret_cast<float4*>(rec + (((2 * oct) ^ (hp & 15)) << 2)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
          *reinterpret_cast<float4*>(rec + (((2 * oct + 1) ^ (hp & 15)) << 2)) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    issue_taps(0, 0, NP);
    __builtin_amdgcn_sched_barrier(0);
    // ---- predictor GEMM of the wave's pixel row: K1p[t*4 + cb] = kernels of channels q*16 + cb*4 .. +3, tap t (16-bit, what
    // the stand-alone F[1] launch would have stored) ----
#ifndef FCVSR_IAC2_KF32
#define FCVSR_IAC2_KF32 1
#endif
    constexpr bool KF = FCVSR_IAC2_KF32 != 0 && NG == 1 && BF;   // kernels kept as (rounded) f32: 24 more registers, 144 fewer conversions per tile (bf16: two VALU ops per pair)
    uint2 K1p[KF ? 1 : 12];
    float K1f[KF ? 12 : 1][4];
    {
      const int x = fresh(), n = x & 15, q = (x & 63) >> 4;
#pragma unroll
      for (int mt = 0; mt < 12; ++mt) {
        const int R = mt * 16 + n;
        const uint4 w0 = *reinterpret_cast<const uint4*>(w_s + R * kJC + ((q ^ (R & 7)) << 3));
        const uint4 w1 = *reinterpret_cast<const uint4*>(w_s + R * kJC + (((q + 4) ^ (R & 7)) << 3));
        f32x4v_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma16x<BF>(w0, kfr[0], acc);
        acc = mfma16x<BF>(w1, kfr[1], acc);
        const float4 b4 = *reinterpret_cast<const float4*>(kb_s + mt * 16 + q * 4);
        const uint2 pk = cvt4<BF>(make_float4(acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w));
        if (KF) cvt16x4_to_f32<BF>(pk, K1f[mt]);
        else K1p[mt] = pk;
        if (mt & 1) __builtin_amdgcn_sched_barrier(0);   // two M-tiles of operands in flight, not all twelve (96 registers)
      }
    }
    // LDS float offsets of the lane's records, once per tile for both directions (16 registers; recomputed per phase they were
    // ~60 VALU instructions per tile of a VALU-bound kernel): ov[cb] = record n, slot (4q + cb) ^ n (vertical reads, v writes: the row is a
    // scalar offset), oh[tt][cb] = record n + tt - 1, slot (4q + cb) ^ ((n + tt - 1) & 15) (horizontal reads)
    int ov[4], oh[3][4];
    {
      const int x = fresh(), n = x & 15, q = (x & 63) >> 4;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        ov[cb] = n * kJC + (((q * 4 + cb) ^ n) << 2);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          const int m = n + tt - 1;
          oh[tt][cb] = m * kJC + (((q * 4 + cb) ^ (m & 15)) << 2);
        }
      }
    }
    auto sac = [&](const int dir) {
      float4 vr[4];
      {
        // ---- vertical: v[row][n] = sum_t s[row + t][n] * K1[row][n][t] ----
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tt = 0; tt < 3; ++tt) {
            const float4 sv = *reinterpret_cast<const float4*>(s_s + (wave + tt) * (kQVX * kJC) + ov[cb]);
            float k[4];
            if (KF) { k[0] = K1f[tt * 4 + cb][0]; k[1] = K1f[tt * 4 + cb][1]; k[2] = K1f[tt * 4 + cb][2]; k[3] = K1f[tt * 4 + cb][3]; }
            else cvt16x4_to_f32<BF>(K1p[tt * 4 + cb], k);
            acc[0] = fmaf(sv.x, k[0], acc[0]); acc[1] = fmaf(sv.y, k[1], acc[1]);
            acc[2] = fmaf(sv.z, k[2], acc[2]); acc[3] = fmaf(sv.w, k[3], acc[3]);
          }
          vr[cb] = make_float4(acc[0], acc[1], acc[2], acc[3]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      issue_fin(dir);                                    // the residual is consumed after the next two barriers
      __syncthreads();                                   // every wave has read its three rows of s: v may overwrite rows 0..3
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) *reinterpret_cast<float4*>(v_s + wave * (kQVX * kJC) + ov[cb]) = vr[cb];
      __syncthreads();
      // ---- horizontal (kernel1 again) + residual + LeakyReLU ----
      const int x = fresh(), n = x & 15, q = (x & 63) >> 4;
      const int gx = tx0 + n - 1;
      if (n >= 1 && n <= kQX && row < H && gx < W) {
        uint16_t* const db = a.dst[dir] + (long long)b * a.dst_sb + (long long)row * a.dst_sy;
        const int e = gx * a.dst_sx + q * 16;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                  // 8 channels at a time
          float f[8], o[8];
          unpack8<ADT>(fpk[hf], f);
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const int cb = hf * 2 + c2;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
              const float4 vv = *reinterpret_cast<const float4*>(v_s + wave * (kQVX * kJC) + oh[tt][cb]);
              float k[4];
              if (KF) { k[0] = K1f[tt * 4 + cb][0]; k[1] = K1f[tt * 4 + cb][1]; k[2] = K1f[tt * 4 + cb][2]; k[3] = K1f[tt * 4 + cb][3]; }
              else cvt16x4_to_f32<BF>(K1p[tt * 4 + cb], k);
              acc[0] = fmaf(vv.x, k[0], acc[0]); acc[1] = fmaf(vv.y, k[1], acc[1]);
              acc[2] = fmaf(vv.z, k[2], acc[2]); acc[3] = fmaf(vv.w, k[3], acc[3]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float r = acc[i] + f[c2 * 4 + i];
              o[c2 * 4 + i] = fmaxf(r, r * slope);       // LeakyReLU for 0 <= slope <= 1 (host-checked): the value of the select form, two ops
            }
          }
          st_p8<ADT>(reinterpret_cast<float*>(db), e + hf * 8, o);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    __builtin_amdgcn_sched_barrier(0);
    warp_store(0);
    __builtin_amdgcn_sched_barrier(0);
    issue_taps(1, 0, EARLY);                  // in flight during direction 0's LDS phases
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    sac(0);
    issue_taps(1, EARLY, NP);
    // the next tile's offsets head its longest dependency chain (offsets -> addresses -> gather): requested here, where few
    // registers are live (before direction 0's phases the two values were spilled, and the spill waited for every gather)
    if (t + 1 < t_end) off_next = load_off(t + 1, fresh());
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                     // v of direction 0 has been read: s may be rewritten
    warp_store(1);
    __syncthreads();
    sac(1);
    __syncthreads();                                     // next tile's warp_store(0) rewrites s / v
  }
}

}  // namespace fcvsr

using namespace fcvsr;

static bool quad_ok(const fcvsr_view* v, int dt) {
  if (!v || !v->ptr || v->dtype != dt || v->sc != 1 || v->c % 4) return false;
  const int g = dt == FCVSR_F32 ? 4 : 4;       // 4-channel accesses: 16 bytes (f32) / 8 bytes (16-bit)
  const int al = dt == FCVSR_F32 ? 16 : 8;
  return v->sx % g == 0 && v->sy % g == 0 && v->sb % g == 0 && ((uintptr_t)v->ptr % al) == 0;
}

static int iac_persistent_wgs() {                      // two workgroups per CU (registers and LDS allow exactly two)
  static int n[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
  if (!n[dev]) {
    hipDeviceProp_t prop;
    n[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? 2 * prop.multiProcessorCount : 512;
  }
  return n[dev];
}

template <int KDT, int ND, bool FK = false>
static void launch_iac64(int adt, dim3 grid, hipStream_t st, IacArgs a) {
  a.ntiles = (int)grid.x;
  // Tile runs pay off for the fused-predictor kernel only (its workgroups stage 24 KB of weights first: 384 -> 350 us per
  // 16 x 180 x 320 iteration); the others lose 3-8 % to a run of any length (measured with 2...28 tiles per workgroup) and
  // keep one tile per workgroup.  FCVSR_IAC_PERSIST: bit 0 = fused-predictor kernel, bit 1 = the others (experiments).
  static const int persist = getenv("FCVSR_IAC_PERSIST") ? atoi(getenv("FCVSR_IAC_PERSIST")) : 1;
  static const int tpw = getenv("FCVSR_IAC_TPW") ? atoi(getenv("FCVSR_IAC_TPW")) : 0;
  if (persist & (FK ? 1 : 2)) {
    int nwg = iac_persistent_wgs() / (int)grid.y;
    if (tpw > 0) nwg = ((int)grid.x + tpw - 1) / tpw;
    nwg = nwg < 8 ? 8 : nwg / 8 * 8;
    if (nwg < (int)grid.x) grid.x = nwg;
  }
  if (adt == FCVSR_F32) hipLaunchKernelGGL((iac_step64_kernel<KDT, FCVSR_F32, ND, FK>), grid, dim3(256), 0, st, a);
  else if (adt == FCVSR_BF16) hipLaunchKernelGGL((iac_step64_kernel<KDT, FCVSR_BF16, ND, FK>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((iac_step64_kernel<KDT, FCVSR_F16, ND, FK>), grid, dim3(256), 0, st, a);
}

template <int KDT>
static void launch_iac(int adt, bool wide, dim3 grid, hipStream_t st, View pv, View ov, View kv, View fv, float slope, int B, int H,
                       int W, View dv, int tx, int ty) {
  if (wide) {
    IacArgs a;
    a.d[0].prev = pv; a.d[0].off = ov; a.d[0].fin = fv; a.d[0].dst = dv;
    a.d[1] = a.d[0];
    a.k1 = kv; a.k0 = kv; a.wk = nullptr; a.kbias = nullptr; a.slope = slope; a.B = B; a.H = H; a.W = W; a.tiles_x = tx; a.tiles_y = ty;
    launch_iac64<KDT, 1>(adt, grid, st, a);
    return;
  }
  if (adt == FCVSR_F32)
    hipLaunchKernelGGL((iac_step_kernel<KDT, FCVSR_F32>), grid, dim3(256), 0, st, pv, ov, kv, fv, slope, B, H, W, dv, tx, ty);
  else if (adt == FCVSR_BF16)
    hipLaunchKernelGGL((iac_step_kernel<KDT, FCVSR_BF16>), grid, dim3(256), 0, st, pv, ov, kv, fv, slope, B, H, W, dv, tx, ty);
  else
    hipLaunchKernelGGL((iac_step_kernel<KDT, FCVSR_F16>), grid, dim3(256), 0, st, pv, ov, kv, fv, slope, B, H, W, dv, tx, ty);
}

extern "C" int fcvsr_iac_step(const fcvsr_view* prev, const fcvsr_view* off, const fcvsr_view* k1, const fcvsr_view* feat_in,
                              float slope, int B, int H, int W, const fcvsr_view* dst, void* stream) {
  FCVSR_CHECK_ARG(prev && prev->ptr, "null prev");
  const int adt = prev->dtype;
  FCVSR_CHECK_ARG(adt == FCVSR_F32 || adt == FCVSR_BF16 || adt == FCVSR_F16, "bad feature dtype");
  FCVSR_CHECK_ARG(quad_ok(prev, adt) && quad_ok(feat_in, adt) && quad_ok(dst, adt),
                  "prev/feat_in/dst: same dtype, channel-contiguous, aligned");
  FCVSR_CHECK_ARG(off && off->ptr && off->c >= 2 && off->dtype == FCVSR_F32, "off needs 2 f32 channels");
  FCVSR_CHECK_ARG(k1 && k1->ptr && k1->sc == 1 && k1->c == 3 * prev->c, "k1 must have 3*C contiguous channels");
  FCVSR_CHECK_ARG(prev->c % kIC == 0 && prev->c == dst->c && prev->c == feat_in->c, "C must be a multiple of 32");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0, "bad sizes");
  const int g = k1->dtype == FCVSR_F32 ? 4 : 8;   // 16-byte (f32) / 8-byte (16-bit) vector loads of 12-element groups
  FCVSR_CHECK_ARG(((uintptr_t)k1->ptr % 16) == 0 && k1->sx % g == 0 && k1->sy % g == 0 && k1->sb % g == 0, "k1 alignment");
  const int tx = cdiv(W, kIX), ty = cdiv(H, kIY);
  // 64-channel kernel: 16-byte accesses of 8 channels (features) / 24 kernel values
  static const bool no_wide = getenv("FCVSR_IAC_WIDE") && atoi(getenv("FCVSR_IAC_WIDE")) == 0;
  const int fa = adt == FCVSR_F32 ? 4 : 8;
  auto wide_ok = [&](const fcvsr_view* v) { return ((uintptr_t)v->ptr % 16) == 0 && v->sx % fa == 0 && v->sy % fa == 0 && v->sb % fa == 0; };
  const bool wide = !no_wide && prev->c % kJC == 0 && wide_ok(prev) && wide_ok(feat_in) && wide_ok(dst) &&
                    (long long)H * prev->sy < (1ll << 31);
  dim3 grid(B * tx * ty, prev->c / (wide ? kJC : kIC));
  hipStream_t st = (hipStream_t)stream;
  const View pv = to_view(*prev), ov = to_view(*off), kv = to_view(*k1), fv = to_view(*feat_in), dv = to_view(*dst);
  if (k1->dtype == FCVSR_F32) launch_iac<FCVSR_F32>(adt, wide, grid, st, pv, ov, kv, fv, slope, B, H, W, dv, tx, ty);
  else if (k1->dtype == FCVSR_BF16) launch_iac<FCVSR_BF16>(adt, wide, grid, st, pv, ov, kv, fv, slope, B, H, W, dv, tx, ty);
  else launch_iac<FCVSR_F16>(adt, wide, grid, st, pv, ov, kv, fv, slope, B, H, W, dv, tx, ty);
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_iac_step2(const fcvsr_view* prev, const fcvsr_view* off, const fcvsr_view* k1, const fcvsr_view* feat_in,
                               float slope, int B, int H, int W, const fcvsr_view* dst, void* stream) {
  FCVSR_CHECK_ARG(prev && off && k1 && feat_in && dst && prev[0].ptr, "null argument");
  const int adt = prev[0].dtype;
  FCVSR_CHECK_ARG(adt == FCVSR_F32 || adt == FCVSR_BF16 || adt == FCVSR_F16, "bad feature dtype");
  const int fa = adt == FCVSR_F32 ? 4 : 8;
  for (int d = 0; d < 2; ++d) {
    FCVSR_CHECK_ARG(quad_ok(&prev[d], adt) && quad_ok(&feat_in[d], adt) && quad_ok(&dst[d], adt),
                    "prev/feat_in/dst: same dtype, channel-contiguous, aligned");
    FCVSR_CHECK_ARG(off[d].ptr && off[d].c >= 2 && off[d].dtype == FCVSR_F32, "off needs 2 f32 channels");
    FCVSR_CHECK_ARG(prev[d].c % kJC == 0 && prev[d].c == prev[0].c && prev[d].c == dst[d].c && prev[d].c == feat_in[d].c,
                    "C must be a multiple of 64 and alike for both directions");
    const fcvsr_view* vs[3] = {&prev[d], &feat_in[d], &dst[d]};
    for (const fcvsr_view* v : vs)
      FCVSR_CHECK_ARG(((uintptr_t)v->ptr % 16) == 0 && v->sx % fa == 0 && v->sy % fa == 0 && v->sb % fa == 0,
                      "16-byte aligned feature views");
  }
  FCVSR_CHECK_ARG(k1->ptr && k1->sc == 1 && k1->c == 3 * prev[0].c, "k1 must have 3*C contiguous channels");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0, "bad sizes");
  const int g = k1->dtype == FCVSR_F32 ? 4 : 8;
  FCVSR_CHECK_ARG(((uintptr_t)k1->ptr % 16) == 0 && k1->sx % g == 0 && k1->sy % g == 0 && k1->sb % g == 0, "k1 alignment");
  const int tx = cdiv(W, kIX), ty = cdiv(H, kIY);
  dim3 grid(B * tx * ty, prev[0].c / kJC);
  IacArgs a;
  for (int d = 0; d < 2; ++d) {
    a.d[d].prev = to_view(prev[d]); a.d[d].off = to_view(off[d]); a.d[d].fin = to_view(feat_in[d]); a.d[d].dst = to_view(dst[d]);
  }
  a.k1 = to_view(*k1); a.k0 = a.k1; a.wk = nullptr; a.kbias = nullptr; a.slope = slope; a.B = B; a.H = H; a.W = W; a.tiles_x = tx; a.tiles_y = ty;
  hipStream_t st = (hipStream_t)stream;
  if (k1->dtype == FCVSR_F32) launch_iac64<FCVSR_F32, 2>(adt, grid, st, a);
  else if (k1->dtype == FCVSR_BF16) launch_iac64<FCVSR_BF16, 2>(adt, grid, st, a);
  else if (k1->dtype == FCVSR_F16) launch_iac64<FCVSR_F16, 2>(adt, grid, st, a);
  else FCVSR_CHECK_ARG(false, "bad k1 dtype");
  FCVSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int fcvsr_iac_step2_fused(const fcvsr_view* prev, const fcvsr_view* off, const fcvsr_view* k0, const void* wk,
                                     const float* kbias, const fcvsr_view* feat_in, float slope, int B, int H, int W,
                                     const fcvsr_view* dst, void* stream) {
  FCVSR_CHECK_ARG(prev && off && k0 && wk && kbias && feat_in && dst && prev[0].ptr, "null argument");
  const int adt = prev[0].dtype;
  FCVSR_CHECK_ARG(adt == FCVSR_F32 || adt == FCVSR_BF16 || adt == FCVSR_F16, "bad feature dtype");
  const int fa = adt == FCVSR_F32 ? 4 : 8;
  for (int d = 0; d < 2; ++d) {
    FCVSR_CHECK_ARG(quad_ok(&prev[d], adt) && quad_ok(&feat_in[d], adt) && quad_ok(&dst[d], adt),
                    "prev/feat_in/dst: same dtype, channel-contiguous, aligned");
    FCVSR_CHECK_ARG(off[d].ptr && off[d].c >= 2 && off[d].dtype == FCVSR_F32, "off needs 2 f32 channels");
    FCVSR_CHECK_ARG(prev[d].c == kJC && dst[d].c == kJC && feat_in[d].c == kJC, "the fused predictor handles C == 64");
    const fcvsr_view* vs[3] = {&prev[d], &feat_in[d], &dst[d]};
    for (const fcvsr_view* v : vs)
      FCVSR_CHECK_ARG(((uintptr_t)v->ptr % 16) == 0 && v->sx % fa == 0 && v->sy % fa == 0 && v->sb % fa == 0,
                      "16-byte aligned feature views");
  }
  FCVSR_CHECK_ARG(k0->ptr && (k0->dtype == FCVSR_BF16 || k0->dtype == FCVSR_F16) && k0->sc == 1 && k0->c == kJC &&
                      ((uintptr_t)k0->ptr % 16) == 0 && k0->sx % 8 == 0 && k0->sy % 8 == 0 && k0->sb % 8 == 0,
                  "k0: 64 contiguous 16-bit channels, 16-byte aligned");
  FCVSR_CHECK_ARG(((uintptr_t)wk % 16) == 0 && ((uintptr_t)kbias % 16) == 0, "weights / bias must be 16-byte aligned");
  FCVSR_CHECK_ARG(B > 0 && H > 0 && W > 0, "bad sizes");
  FCVSR_CHECK_ARG((long long)H * prev[0].sy < (1ll << 31) && (long long)H * prev[1].sy < (1ll << 31), "image too large");
  const int tx = cdiv(W, kIX), ty = cdiv(H, kIY);
  dim3 grid(B * tx * ty, 1);
  IacArgs a;
  for (int d = 0; d < 2; ++d) {
    a.d[d].prev = to_view(prev[d]); a.d[d].off = to_view(off[d]); a.d[d].fin = to_view(feat_in[d]); a.d[d].dst = to_view(dst[d]);
  }
  a.k1 = to_view(*k0); a.k0 = to_view(*k0); a.wk = (const uint16_t*)wk; a.kbias = kbias;
  a.slope = slope; a.B = B; a.H = H; a.W = W; a.tiles_x = tx; a.tiles_y = ty;
  hipStream_t st = (hipStream_t)stream;
  static const int form = getenv("FCVSR_IAC_FORM") ? atoi(getenv("FCVSR_IAC_FORM")) : 2;   // 1 = iac_step64_kernel (kernels through LDS)
  bool same = adt != FCVSR_F32 && adt == k0->dtype && (long long)H * k0->sy < (1ll << 31);
  {
    const fcvsr_view* sets[4] = {prev, off, feat_in, dst};
    for (const fcvsr_view* v : sets)
      same = same && v[0].sb == v[1].sb && v[0].sy == v[1].sy && v[0].sx == v[1].sx && v[0].sc == v[1].sc && (long long)H * v[0].sy < (1ll << 31);
  }
  if (form == 2 && same && slope >= 0.f && slope <= 1.f) {   // kernels in registers, three workgroups per CU (iac_fused2_kernel)
    Iac2Args q;
    for (int d = 0; d < 2; ++d) {
      q.prev[d] = (const uint16_t*)prev[d].ptr; q.off[d] = (const float*)off[d].ptr;
      q.fin[d] = (const uint16_t*)feat_in[d].ptr; q.dst[d] = (uint16_t*)dst[d].ptr;
    }
    q.k0 = (const uint16_t*)k0->ptr; q.wk = (const uint16_t*)wk; q.kbias = kbias;
    q.prev_sb = prev[0].sb; q.off_sb = off[0].sb; q.fin_sb = feat_in[0].sb; q.dst_sb = dst[0].sb; q.k0_sb = k0->sb;
    q.prev_sy = (int)prev[0].sy; q.prev_sx = (int)prev[0].sx; q.off_sy = (int)off[0].sy; q.off_sx = (int)off[0].sx; q.off_sc = (int)off[0].sc;
    q.fin_sy = (int)feat_in[0].sy; q.fin_sx = (int)feat_in[0].sx; q.dst_sy = (int)dst[0].sy; q.dst_sx = (int)dst[0].sx;
    q.k0_sy = (int)k0->sy; q.k0_sx = (int)k0->sx;
    q.slope = slope; q.H = H; q.W = W; q.tiles_x = cdiv(W, kQX); q.tiles_y = cdiv(H, kQY); q.ntiles = B * q.tiles_x * q.tiles_y;
    int nwg = iac_persistent_wgs() / 2 * FCVSR_IAC2_WGS;
    nwg = nwg < 8 ? 8 : nwg / 8 * 8;
    if (nwg > q.ntiles) nwg = q.ntiles;
    if (adt == FCVSR_BF16) hipLaunchKernelGGL((iac_fused2_kernel<FCVSR_BF16, FCVSR_BF16>), dim3(nwg), dim3(256), 0, st, q);
    else hipLaunchKernelGGL((iac_fused2_kernel<FCVSR_F16, FCVSR_F16>), dim3(nwg), dim3(256), 0, st, q);
    FCVSR_LAUNCH_CHECK();
    return 0;
  }
  if (k0->dtype == FCVSR_BF16) launch_iac64<FCVSR_BF16, 2, true>(adt, grid, st, a);
  else launch_iac64<FCVSR_F16, 2, true>(adt, grid, st, a);
  FCVSR_LAUNCH_CHECK();
  return 0;
}
