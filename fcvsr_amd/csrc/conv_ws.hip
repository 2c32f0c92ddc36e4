// Weight-stationary 3x3 implicit-GEMM convolution for the 64-input-channel layers that carry most of the FLOPs of the path
// (BlockRCB / RCB bodies, conv_KP, F.0, recorb0: reference CVSR_freq.py:712-777, :1395-1410, :2555-2560).
//
// conv3_lean_kernel re-stages the 72 KiB weight block through LDS for every 4 x 32 pixel tile and feeds BOTH MFMA operands
// from LDS (3 ds_read_b128 per 2 MFMAs, two barriers per tap).  Here the weights never move after the prologue:
//   * persistent workgroups (one per CU, 8 waves); a wave owns one 32-cout slice ("role") and keeps its whole
//     B operand - 9 taps x 4 k-steps x 16 bytes per lane = 144 VGPRs - in registers for the lifetime of the kernel;
//   * the only LDS traffic of the main loop is the A operand: one ds_read_b128 per MFMA, no barrier inside a tile;
//   * the 10 x 34 pixel halo tile of the NEXT 8 x 32 tile is copied global -> LDS by LDS-DMA (global_load_lds_dwordx4,
//     no staging registers) while the current tile is multiplied: one barrier per tile;
//   * LDS-DMA writes 64 lanes x 16 bytes linearly, so the image cannot be padded; bank conflicts are avoided by an XOR
//     swizzle applied on the per-lane SOURCE address: 16-byte chunk c of halo pixel p lands in slot c ^ ((p >> 1) & 7),
//     which makes the 16 lanes of every ds_read_b128 phase hit 16 distinct 16-byte bank groups;
//   * tiles are handed out so that the 32 workgroups of one XCD walk a contiguous range of tiles (halo rows hit its L2).
// The epilogue (bias, activation, up to two residuals, 16-bit or f32 store) is per wave through a private LDS transpose.
#include <stdlib.h>
#include "conv_ws.h"
#include "mfma_util.h"

namespace fcvsr {

constexpr int kWsNW = 4;                           // waves per workgroup (two workgroups per CU run out of phase)
constexpr int kWsTH = 4, kWsTW = 32, kWsHW = kWsTW + 2, kWsHH = kWsTH + 2;
constexpr int kWsNHP = kWsHH * kWsHW;              // 204 halo pixels of 64 channels = 128 bytes each
constexpr int kWsNG = (kWsNHP + 7) / 8;            // 26 LDS-DMA wave-instructions (8 pixels = 1 KiB each)
constexpr int kWsABytes = kWsNG * 1024;
constexpr int kWsERow = 36;                        // floats per pixel row of the epilogue transpose (32 couts + pad)
constexpr int kWsEBytes = 32 * kWsERow * 4;

struct WsTile {
  int gi, b, ty0, tx0;
};

__device__ __forceinline__ WsTile ws_decode(const WsArgs& a, int tile) {
  WsTile t;
  t.gi = 0;
  if (a.n_groups > 1 && tile >= a.g[1].tile_begin) t.gi = 1;
  if (a.n_groups > 2 && tile >= a.g[2].tile_begin) t.gi = 2;
  const WsGroup& G = a.g[t.gi];
  const int tl = tile - G.tile_begin;
  const int per_img = G.tiles_x * G.tiles_y;
  t.b = tl / per_img;
  const int t2 = tl - t.b * per_img;
  const int ty = t2 / G.tiles_x;
  t.ty0 = ty * kWsTH;
  t.tx0 = (t2 - ty * G.tiles_x) * kWsTW;
  return t;
}

// global -> LDS copy of the halo tile (all 8 waves, 5-6 wave-instructions each).  Offsets are 32-bit (the dispatcher checks
// that every source spans < 2^29 elements) and the per-lane products fit 24-bit multiplies.
__device__ __forceinline__ void ws_stage(const WsArgs& a, const WsTile& t, unsigned char* abuf, int wave, int lane) {
  const WsGroup& G = a.g[t.gi];
  const char* sbase = reinterpret_cast<const char*>(G.src.p) + (long long)t.b * G.src.sb * 2;
  const int sy2 = (int)G.src.sy * 2, sx2 = (int)G.src.sx * 2;
  const int H = G.H, W = G.W;
  const int tile_off = (t.ty0 - 1) * sy2 + (t.tx0 - 1) * sx2;          // wave-uniform
  const int sub = lane >> 3, cl = lane & 7;
#pragma unroll
  for (int i = 0; i < (kWsNG + kWsNW - 1) / kWsNW; ++i) {
    const int g = wave + kWsNW * i;
    if (g < kWsNG) {
      const int p = g * 8 + sub;
      const int hy = __mul24(p, 241) >> 13;         // p / 34 for p < 344
      const int hx = p - __mul24(hy, kWsHW);
      const int iy = t.ty0 - 1 + hy, ix = t.tx0 - 1 + hx;
      const int c = cl ^ ((p >> 1) & 7);
      const bool ok = (p < kWsNHP) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      const int off = tile_off + __mul24(hy, sy2) + __mul24(hx, sx2) + c * 16;
      const char* src = ok ? sbase + off : reinterpret_cast<const char*>(a.zeros);
      // Issued as inline asm on purpose: with the builtin the compiler treats every later LDS read (the A fragments of the
      // CURRENT tile, the epilogue transposes) as dependent on the copy and drains it with vmcnt(0) first, which serialises
      // the copy of tile t+1 with the multiplication of tile t.  The kernel waits for the copy itself (see the tile loop).
      const unsigned lds_off = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(abuf + g * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(lds_off) : "memory");
    }
  }
}

template <bool BF16, bool DST16, int R>
__global__ __launch_bounds__(kWsNW * 64, 2) void conv3_ws_kernel(WsArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int RW = R < kWsNW ? R : kWsNW;          // roles (32-cout slices) inside one workgroup
  constexpr int NB = R / RW;                         // cout blocks: workgroups alternate between them and keep theirs for good
  constexpr int RG = kWsNW / RW;                     // row groups
  constexpr int JW = kWsTH / RG;                     // rows of the tile computed by one wave
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int rg = wave / RW;
  float* E_s = reinterpret_cast<float*>(lds + 2 * kWsABytes + wave * kWsEBytes);

  // ---- tile schedule: the workgroups of one XCD (blockIdx % 8) share a contiguous range of tiles -----------------------
  const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) / NB, nslots = (gridDim.x >> 3) / NB;
  const int role = ((blockIdx.x >> 3) % NB) * RW + wave % RW;
  const int tb = (int)((long long)xcd * a.total_tiles / 8), te = (int)((long long)(xcd + 1) * a.total_tiles / 8);
  int tile = tb + slot;
  if (tile >= te) return;                            // uniform per workgroup

  // ---- stationary B operand: this wave's 32 couts x (9 taps x 64 cin) ------------------------------------------------
  uint4 wreg[36];
  {
    const uint16_t* wp = a.w + (long long)(role * 32 + r) * a.cin_pad + h * 8;
    const long long wtap = (long long)a.cout_pad * a.cin_pad;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wreg[tap * 4 + kk] = *reinterpret_cast<const uint4*>(wp + tap * wtap + kk * 16);
  }

  WsTile cur = ws_decode(a, tile);
  ws_stage(a, cur, lds, wave, lane);
  int buf = 0;

  float slope = a.slope;
  if (a.act == FCVSR_ACT_PRELU) slope = a.slope_ptr[0];
  const int act = a.act;
  const bool r16 = a.res16 != 0;
  // per-lane epilogue constants (the role is fixed, so the lane's output channels never change): loaded once, because an
  // ordinary global load inside the tile loop would make the compiler drain the LDS-DMA queue (vmcnt(0)) at its first use
  const int en = role * 32 + (DST16 ? (lane & 3) * 8 : (lane & 7) * 4);
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (a.bias) {
    b0 = *reinterpret_cast<const float4*>(a.bias + en);
    if (DST16) b1 = *reinterpret_cast<const float4*>(a.bias + en + 4);
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // the first tile has landed
  while (true) {
    const int next = tile + nslots;
    WsTile nxt = cur;
    if (next < te) {
      nxt = ws_decode(a, next);
      if (!(a.dbg & 1)) ws_stage(a, nxt, lds + (buf ^ 1) * kWsABytes, wave, lane);
    }
    const WsGroup& G = a.g[cur.gi];
    const unsigned char* A_s = lds + buf * kWsABytes;
    const int H = G.H, W = G.W;
    const View dv = G.dst, r0v = G.res[0], r1v = G.res[1];

#pragma unroll 1
    for (int j = 0; j < JW; ++j) {
      const int row = rg + j * RG;
      const int py = cur.ty0 + row;
      const bool live = py < H;                      // wave-uniform
      f32x16_t acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const int pb = row * kWsHW + r;
      if (live && !(a.dbg & 2)) {
      uint4 af[2][4];                                // A fragments: one tap (4 k-steps) ahead of the MFMAs
#define FCVSR_WS_LOAD_A(TAP, SLOT)                                                                     \
  do {                                                                                                 \
    const int pix_ = pb + ((TAP) / 3) * kWsHW + ((TAP) % 3);                                            \
    const int a0_ = pix_ * 128 + ((h ^ ((pix_ >> 1) & 7)) << 4);                                        \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                    \
        af[SLOT][kk] = *reinterpret_cast<const uint4*>(A_s + (a0_ ^ (kk * 32)));                        \
  } while (0)
      FCVSR_WS_LOAD_A(0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) FCVSR_WS_LOAD_A(tap + 1, (tap + 1) & 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = mfma<BF16>(af[tap & 1][kk], wreg[tap * 4 + kk], acc);
        // keep the schedule of the source: the next tap's 4 LDS reads are issued ahead of this tap's 4 MFMAs (left alone,
        // the scheduler sinks every read next to its use - one read in flight - and the MFMA pipe waits on LDS latency)
        if (tap + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
#undef FCVSR_WS_LOAD_A
      }
      // The next tile's LDS-DMA (issued at the top of this tile) must have landed before the barrier below.  Waiting HERE,
      // ahead of the last row's stores, keeps those stores out of the wait: vmcnt counts stores too, and a wait placed
      // after them would expose a full store round trip per tile.
      if (j == JW - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!live) continue;

      // ---- epilogue: 32 pixels of row py x this role's 32 couts ---------------------------------------------------------
#pragma unroll
      for (int i = 0; i < 16; ++i) E_s[((i & 3) + 8 * (i >> 2) + 4 * h) * kWsERow + r] = acc[i];
      __builtin_amdgcn_wave_barrier();
      const long long drow = (long long)cur.b * dv.sb + (long long)py * dv.sy;
      const long long r0off = (long long)cur.b * r0v.sb + (long long)py * r0v.sy;
      const long long r1off = (long long)cur.b * r1v.sb + (long long)py * r1v.sy;
      if (DST16) {
        const int co = lane & 3, psub = lane >> 2;   // 8 couts per lane -> one 16-byte store; 16 pixels per pass
        const int n = en;
        uint16_t* dp = reinterpret_cast<uint16_t*>(dv.p) + drow + n;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int m = q * 16 + psub;
          const int px = cur.tx0 + m;
          if (px < W) {
            const float* es = E_s + m * kWsERow + co * 8;
            const float4 va = *reinterpret_cast<const float4*>(es), vb = *reinterpret_cast<const float4*>(es + 4);
            float x[8] = {va.x + b0.x, va.y + b0.y, va.z + b0.z, va.w + b0.w, vb.x + b1.x, vb.y + b1.y, vb.z + b1.z, vb.w + b1.w};
            if (act == FCVSR_ACT_RELU) {
#pragma unroll
              for (int k = 0; k < 8; ++k) x[k] = fmaxf(x[k], 0.f);
            } else if (act != FCVSR_ACT_NONE) {
#pragma unroll
              for (int k = 0; k < 8; ++k) x[k] = x[k] >= 0.f ? x[k] : x[k] * slope;
            }
            if (a.n_res > 0) {
              float rr[8];
              load_res<BF16, 8>(r0v.p, r0off + (long long)px * r0v.sx + n, r16, rr);
#pragma unroll
              for (int k = 0; k < 8; ++k) x[k] = fmaf(a.rs[0], rr[k], x[k]);
            }
            if (a.n_res > 1) {
              float rr[8];
              load_res<BF16, 8>(r1v.p, r1off + (long long)px * r1v.sx + n, r16, rr);
#pragma unroll
              for (int k = 0; k < 8; ++k) x[k] = fmaf(a.rs[1], rr[k], x[k]);
            }
            const uint2 lo = cvt4<BF16>(make_float4(x[0], x[1], x[2], x[3])), hi = cvt4<BF16>(make_float4(x[4], x[5], x[6], x[7]));
            if (!(a.dbg & 4) || x[0] == 12345.678f) *reinterpret_cast<uint4*>(dp + (long long)px * dv.sx) = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
        }
      } else {
        const int cq = lane & 7, psub = lane >> 3;   // 4 couts per lane -> one 16-byte store; 8 pixels per pass
        const int n = en;
        float* dp = dv.p + drow + n;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = q * 8 + psub;
          const int px = cur.tx0 + m;
          if (px < W) {
            const float4 va = *reinterpret_cast<const float4*>(E_s + m * kWsERow + cq * 4);
            float x[4] = {va.x + b0.x, va.y + b0.y, va.z + b0.z, va.w + b0.w};
            if (act == FCVSR_ACT_RELU) {
#pragma unroll
              for (int k = 0; k < 4; ++k) x[k] = fmaxf(x[k], 0.f);
            } else if (act != FCVSR_ACT_NONE) {
#pragma unroll
              for (int k = 0; k < 4; ++k) x[k] = x[k] >= 0.f ? x[k] : x[k] * slope;
            }
            if (a.n_res > 0) {
              float rr[4];
              load_res<BF16, 4>(r0v.p, r0off + (long long)px * r0v.sx + n, r16, rr);
#pragma unroll
              for (int k = 0; k < 4; ++k) x[k] = fmaf(a.rs[0], rr[k], x[k]);
            }
            if (a.n_res > 1) {
              float rr[4];
              load_res<BF16, 4>(r1v.p, r1off + (long long)px * r1v.sx + n, r16, rr);
#pragma unroll
              for (int k = 0; k < 4; ++k) x[k] = fmaf(a.rs[1], rr[k], x[k]);
            }
            *reinterpret_cast<float4*>(dp + (long long)px * dv.sx) = make_float4(x[0], x[1], x[2], x[3]);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();               // E_s is rewritten by the next row
    }

    if (next >= te) break;
    // raw barrier: __syncthreads() would add a vmcnt(0) for the stores just issued.  LDS reads of this tile are complete
    // (their MFMAs have issued); the E_s transposes are private to each wave.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    tile = next;
    cur = nxt;
    buf ^= 1;
  }
}

bool conv3_ws_supports(int cin, int cout) { return cin == 64 && (cout == 64 || cout == 128 || cout == 256); }

template <bool BF16, bool DST16, int R>
static hipError_t launch_ws(const WsArgs& a, hipStream_t st) {
  static int n_cu = 0;
  constexpr size_t lds = 2ull * kWsABytes + (size_t)kWsNW * kWsEBytes;
  constexpr int NB = R > kWsNW ? R / kWsNW : 1;
  if (!n_cu) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv3_ws_kernel<BF16, DST16, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    n_cu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
  }
  // two persistent workgroups per CU (they drift out of phase: one multiplies while the other stores), a multiple of 8*NB
  int grid = 2 * n_cu;
  const int need = (a.total_tiles + 7) / 8 * 8 * NB;
  if (grid > need) grid = need;
  hipLaunchKernelGGL((conv3_ws_kernel<BF16, DST16, R>), dim3(grid), dim3(kWsNW * 64), lds, st, a);
  return hipGetLastError();
}

template <bool BF16, bool DST16>
static hipError_t launch_ws_r(const WsArgs& a, hipStream_t st) {
  if (a.cout == 64) return launch_ws<BF16, DST16, 2>(a, st);
  if (a.cout == 128) return launch_ws<BF16, DST16, 4>(a, st);
  if (a.cout == 256) return launch_ws<BF16, DST16, 8>(a, st);
  return hipErrorInvalidValue;
}

hipError_t launch_conv3_ws(const WsArgs& a, bool bf16, bool dst16, hipStream_t st) {
  if (!conv3_ws_supports(a.cin, a.cout)) return hipErrorInvalidValue;
  if (bf16) return dst16 ? launch_ws_r<true, true>(a, st) : launch_ws_r<true, false>(a, st);
  return dst16 ? launch_ws_r<false, true>(a, st) : launch_ws_r<false, false>(a, st);
}

}  // namespace fcvsr
