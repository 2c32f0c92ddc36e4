// 3x3 implicit-GEMM convolution with LDS-RESIDENT weights for the 64 / 128-channel layers that carry the FLOPs of the path
// (BlockRCB / RCB bodies, conv_KP, F.0, group convs, recorb0, conv3: reference CVSR_freq.py:705-803, :1409-1416, :1430, :2608).
//
// conv3_lean_kernel (conv_mfma.hip) re-stages the 72 KiB weight block through LDS for every 4 x 32 pixel tile, tap by tap behind
// two barriers per tap, and a wave owns 1 x 2 MFMA fragments (1.5 ds_read_b128 per MFMA): counting the weight re-staging the
// LDS pipe is oversubscribed, which is what held it at 30 % of the MFMA roof.  Here:
//   * persistent workgroups, one per CU (512 threads = 8 waves, two per SIMD); the 9 taps x 64 cin x 64 cout weight block
//     (73,728 bytes, or 2 cin chunks x 9 taps x 32 couts for the 128-input-channel layers) is copied into LDS ONCE per workgroup;
//   * the workgroup walks 8 x 32 pixel tiles.  Its two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) alternate roles
//     every phase: one group multiplies its tile (2 rows x all couts per wave = 2 x 2 fragments, 1.0 ds_read_b128 per MFMA)
//     while the other stores its previous tile straight from the accumulators and copies its next 10 x 34 halo tile into its
//     own LDS buffer by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write).  ONE barrier per phase;
//   * operand roles are swapped (weights = MFMA A operand, pixels = B operand), so a lane ends up with 4 consecutive output
//     channels of ONE pixel per accumulator quad; a v_permlane32_swap pairs the quads of the two half-waves into 8 consecutive
//     channels = one 16-byte store (16-bit destination).  No LDS transpose, hence no epilogue barrier and no epilogue LDS;
//   * LDS-DMA writes 64 lanes x 16 bytes linearly, so rows cannot be padded: bank conflicts are avoided by an XOR swizzle on the
//     per-lane SOURCE address (16-byte chunk c of halo column hx lands in slot c ^ ((hx >> 1) & 7); weights: by cout), which
//     makes the 16 lanes of every ds_read_b128 phase hit 16 distinct 16-byte bank groups;
//   * the 32 workgroups of one XCD (blockIdx % 8) share a contiguous range of tiles, so halo rows and both cout blocks of a
//     tile hit in that XCD's L2.
// LDS: 73,728 (weights) + 2 x 44,032 (halo tiles) + 256 (bias) + 384 (group parameters) = 162,432 of 163,840 bytes.
#include <stdlib.h>
#include <type_traits>
#include "conv_res.h"
#include "mfma_util.h"

namespace fcvsr {

constexpr int kRTH = 8, kRTW = 32, kRHW = kRTW + 2, kRHH = kRTH + 2;
constexpr int kRNHP = kRHH * kRHW;                 // 340 halo pixels of 64 channels = 128 bytes each
constexpr int kRNG = (kRNHP + 7) / 8;              // 43 LDS-DMA wave-instructions (8 pixels = 1 KiB each)
constexpr int kRXBytes = kRNG * 1024;              // 44,032
constexpr int kRWRows = 576;                       // weight rows of 64 cin (128 bytes)
constexpr int kRWBytes = kRWRows * 128;            // 73,728
constexpr int kRRowB = kRHW * 128;                 // bytes per halo row
constexpr int kRBiasOff = kRWBytes + 2 * kRXBytes;  // 64 bias floats
constexpr int kRTabOff = kRBiasOff + 256;           // per-group parameter table: 3 x 32 dwords
constexpr size_t kRLds = (size_t)kRTabOff + 3 * 32 * 4;

// Per-group parameters as 32-bit words.  The kernel reads them from LDS: indexing the kernel-argument block by a run-time group
// index makes hipcc fetch it with VECTOR loads, whose s_waitcnt vmcnt(0) also waits for the stores of the previous tile that the
// storing role leaves in flight on purpose (measured: 5-6 k of a 9 k-cycle phase).  Offsets are 32-bit: the dispatcher checks
// that every tensor spans < 2^29 elements.
enum {
  kTSrcLo = 0, kTSrcHi, kTSrcSb2, kTSrcSy2, kTSrcSx2, kTH, kTW, kTTilesX, kTPerImg, kTTileBegin,
  kTDstLo = 12, kTDstHi, kTDstSb, kTDstSy, kTDstSx, kTR0Lo, kTR0Hi, kTR0Sb, kTR0Sy, kTR0Sx, kTR1Lo, kTR1Hi, kTR1Sb, kTR1Sy, kTR1Sx
};

struct ResK {                                        // kernel arguments
  int n_groups, total_tiles, cout, cout_pad, cin_pad, act, n_res, res16, dbg, ps;
  float slope, rs[2];
  const float* slope_ptr;
  const uint16_t* w;
  const float* bias;
  const void* zeros;
  unsigned long long* stamps;
  int tab[3][32];
};

struct ResTile {
  int gi, b, ty0, tx0;
};

__device__ __forceinline__ ResTile res_decode(const int* tabL, int n_groups, int tile) {
  ResTile t;
  t.gi = 0;
  if (n_groups > 1 && tile >= tabL[32 + kTTileBegin]) t.gi = 1;
  if (n_groups > 2 && tile >= tabL[64 + kTTileBegin]) t.gi = 2;
  const int* T = tabL + t.gi * 32;
  const int tl = tile - T[kTTileBegin];
  const int per_img = T[kTPerImg], tiles_x = T[kTTilesX];
  t.b = tl / per_img;
  const int t2 = tl - t.b * per_img;
  const int ty = t2 / tiles_x;
  t.ty0 = ty * kRTH;
  t.tx0 = (t2 - ty * tiles_x) * kRTW;
  return t;
}

// Pointers rebuilt from table words are declared GLOBAL (address space 1): a generic pointer makes hipcc emit flat_load /
// flat_store, which count on lgkmcnt as well and turn every counted LDS wait of the multiply loop into lgkmcnt(0).
typedef __attribute__((address_space(1))) char gchar_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;      // plain vectors: HIP's uint4 / float4 classes have no
typedef __attribute__((ext_vector_type(4))) float f32x4_t;         // address-space-qualified assignment operators
typedef __attribute__((address_space(1))) u32x4_t guint4_t;
typedef __attribute__((address_space(1))) f32x4_t gfloat4_t;
__device__ __forceinline__ void gstore(gchar_t* p, uint4 v) { *reinterpret_cast<guint4_t*>(p) = u32x4_t{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void gstore(gchar_t* p, float4 v) { *reinterpret_cast<gfloat4_t*>(p) = f32x4_t{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ uint4 gload_u4(const gchar_t* p) { const u32x4_t v = *reinterpret_cast<const guint4_t*>(p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 gload_f4(const gchar_t* p) { const f32x4_t v = *reinterpret_cast<const gfloat4_t*>(p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ gchar_t* tab_ptr(const int* T, int lo) {
  return reinterpret_cast<gchar_t*>(((unsigned long long)(unsigned)T[lo + 1] << 32) | (unsigned)T[lo]);
}

// One LDS-DMA wave-instruction: lane l copies 16 bytes from its own global address to LDS byte (lds_off + 16 l).
// Inline asm on purpose: with the builtin the compiler orders every later LDS read behind the copy (vmcnt(0)), which would
// serialise the copy of the next tile with the multiplication of the current one.  The kernel waits for its copies itself.
__device__ __forceinline__ void glds16(const void* src, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_off) : "memory");
}

// global -> LDS copy of the 64-channel chunk `ch` of a halo tile by the 4 waves of one group (wq = 0..3), 10-11 wave-instructions
// each, straight-line (halo pixels outside the image, and the 4 pixels past the tile in the last instruction, read the zero page).
__device__ __forceinline__ void res_stage(const int* tabL, const void* zeros, const ResTile& t, int ch, unsigned xoff, int wq, int lane) {
  const int* T = tabL + t.gi * 32;
  const char* sbase = (const char*)(tab_ptr(T, kTSrcLo) + (long long)t.b * T[kTSrcSb2] + ch * 128);
  const int sy2 = T[kTSrcSy2], sx2 = T[kTSrcSx2];
  const int H = T[kTH], W = T[kTW];
  const int sub = lane >> 3, cl = lane & 7;
#pragma unroll
  for (int i = 0; i < (kRNG + 3) / 4; ++i) {
    const int g = wq + 4 * i;
    if (i * 4 + 3 < kRNG || g < kRNG) {              // only the last round (i = 10) has a (wave-uniform) condition
      const int p = g * 8 + sub;
      const int hy = __mul24(p, 241) >> 13;         // p / 34 for p < 344
      const int hx = p - __mul24(hy, kRHW);
      const int iy = t.ty0 - 1 + hy, ix = t.tx0 - 1 + hx;
      const int c = cl ^ ((hx >> 1) & 7);
      const bool ok = (p < kRNHP) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      const int off = __mul24(iy, sy2) + __mul24(ix, sx2) + c * 16;
      const char* src = ok ? sbase + off : reinterpret_cast<const char*>(zeros);
      glds16(src, __builtin_amdgcn_readfirstlane(xoff + g * 1024));
    }
  }
}

// Diagnostic build switch only (a.stamps != nullptr): in-kernel cycle stamps of one workgroup, written to a buffer nothing else
// reads (scripts/res_stamps.py prints where a phase spends its cycles).  No stamp executes in the normal path.
#define FCVSR_RES_STAMP(SLOT)                                                                                          \
  do {                                                                                                                 \
    if (a.stamps && blockIdx.x == 8 && p < 64 && lane == 0)                                                            \
      a.stamps[(wave * 64 + p) * 8 + (SLOT)] = __builtin_amdgcn_s_memtime();                                           \
  } while (0)

// NCH = input-channel chunks of 64 (1 or 2); a workgroup owns CO = 64 / NCH output channels of every tile it visits.
// MODE: 0 = f32 destination, 1 = 16-bit destination, 2 = 16-bit destination without residuals (the multiplying wave packs).
// NSU: 1 = the activation's negative-side factor lies in [0, 1] (ReLU, LeakyReLU): act(x) = max(x, ns * x); 2 = no activation at
// all (half of the BlockRCB layers): the multiply + max per value is skipped, same bits as max(x, 1 * x); 0 = general (PReLU).
// Both are template parameters because hipcc re-merges wave-uniform run-time variants into one body with a scalar branch
// per 4 values (SimplifyCFG hoists the common code of the arms): ~40 branches per tile in the epilogue.
template <bool BF16, int MODE, int NCH, int NSU>
__global__ __launch_bounds__(512, 2) void conv3_res_kernel(ResK a) {
  constexpr bool DST16 = MODE != 0;
  constexpr bool FAST = MODE == 2;
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int CO = 64 / NCH;                       // couts per workgroup
  constexpr int MF = CO / 32;                        // weight (A operand) fragments per wave
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: tile decoding and role branches stay on the SALU
  const int r = lane & 31, h = lane >> 5;
  const int grp = wave >> 2, wq = wave & 3;          // role group (0: waves 0-3, 1: waves 4-7), row pair inside the tile
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

  // ---- schedule: the workgroups of one XCD (blockIdx % 8) share a contiguous range of tiles; NB cout blocks per tile ------
  const int NB = a.cout / CO;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int nb = loc % NB, slot = loc / NB, nslots = (gridDim.x >> 3) / NB;
  const int tb = (int)((long long)xcd * a.total_tiles / 8), te = (int)((long long)(xcd + 1) * a.total_tiles / 8);
  const int n0 = nb * CO;
  const int ntile = (tb + slot < te) ? (te - tb - slot + nslots - 1) / nslots : 0;     // tiles of this workgroup
  if (ntile == 0) return;                                                               // uniform per workgroup
  const int nmine = (ntile - grp + 1) >> 1;          // tiles of my group: list indices grp, grp + 2, ...
  const int NU = nmine * NCH;                        // my units (tile, chunk)
  const int plast = 2 * ((ntile + 1) >> 1) * NCH;    // last phase (group 0's final epilogue / group 1's last compute + 1)

  // ---- resident weights: rows [(ch*9 + tap) * CO + co] of 64 cin, chunk c of a row in slot c ^ ((co >> 1) & 7) -----------
  {
    const int sub = lane >> 3, cl = lane & 7;
#pragma unroll
    for (int i = 0; i < kRWRows / 8 / 8; ++i) {
      const int g = wave + 8 * i;
      const int row = g * 8 + sub;
      const int q = row / CO, co = row - q * CO;     // CO is a power of two
      const int ch = q / 9, tap = q - ch * 9;
      const int c = cl ^ ((co >> 1) & 7);
      const uint16_t* src = a.w + ((long long)tap * a.cout_pad + n0 + co) * a.cin_pad + ch * 64 + c * 8;
      glds16(src, __builtin_amdgcn_readfirstlane(lds0 + g * 1024));
    }
  }
  if (tid < CO) reinterpret_cast<float*>(lds + kRBiasOff)[tid] = a.bias ? a.bias[n0 + tid] : 0.f;
  if (tid >= 128 && tid < 128 + 96) reinterpret_cast<int*>(lds + kRTabOff)[tid - 128] = a.tab[(tid - 128) >> 5][(tid - 128) & 31];
  const int* tabL = reinterpret_cast<const int*>(lds + kRTabOff);
  const unsigned xoff = lds0 + kRWBytes + grp * kRXBytes;                // my group's halo buffer
  __syncthreads();                                   // the parameter table is in LDS
  ResTile tcur = res_decode(tabL, a.n_groups, tb + slot + grp * nslots < te ? tb + slot + grp * nslots : tb + slot);   // the tile my group staged last
  if (grp == 0 && !(a.dbg & 1)) res_stage(tabL, a.zeros, tcur, 0, xoff, wq, lane);

  // Activation as max(x, ns * x) (0 <= ns <= 1) or max(x, 0) + ns * min(x, 0) with ns = 0 (ReLU), slope (LeakyReLU / PReLU) or 1 (none): the same values as the
  // branchy form, and no per-element scalar branch on `act` (hipcc does not unswitch it: 64 branches per tile row).
  // The PReLU slope is read by an explicitly GLOBAL load: a flat_load (address space not provable) stays "pending" in the
  // compiler's wait-count bookkeeping for the rest of the kernel and turns every counted lgkmcnt(N) of the multiply loop into
  // lgkmcnt(0).
  float ns = 1.f;
  if (a.act == FCVSR_ACT_RELU) ns = 0.f;
  else if (a.act == FCVSR_ACT_LEAKY) ns = a.slope;
  else if (a.act == FCVSR_ACT_PRELU) ns = *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(a.slope_ptr));
  const bool r16 = a.res16 != 0;

  // ---- fragment addresses (LDS byte offsets relative to lds) ------------------------------------------------------------
  unsigned wb[4], xb[3][4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    wb[kk] = r * 128 + (((2 * kk + h) ^ ((r >> 1) & 7)) << 4);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      xb[kx][kk] = kRWBytes + grp * kRXBytes + (2 * wq * kRHW + r + kx) * 128 + (((2 * kk + h) ^ (((r + kx) >> 1) & 7)) << 4);
  }

  f32x16_t acc[MF][2];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // weights and group 0's first tile have landed

  int k = 0;                                         // my next unit to multiply
#pragma unroll 1
  for (int p = 0;; ++p) {
    FCVSR_RES_STAMP(0);
    if ((p & 1) == grp) {
      // ================= multiply unit k: tile list index 2 * (k / NCH) + grp, chunk k % NCH ==========================
      if (k < NU) {
        const int ch = (NCH == 1) ? 0 : (k & (NCH - 1));
        if (ch == 0) {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int i = 0; i < 16; ++i) acc[mf][j][i] = 0.f;
        }
        if (!(a.dbg & 2)) {
          __builtin_amdgcn_s_setprio(2);   // the multiplying wave wins VALU/MFMA issue over its SIMD partner's epilogue
          const unsigned wch = ch * (9 * CO * 128);
          uint4 wf[3][MF], xf[3][2];                   // fragments are read two 16-deep steps ahead of their MFMAs
#define FCVSR_RES_LOAD(S, SLOT)                                                                                  \
  do {                                                                                                           \
    constexpr int tap_ = (S) / 4, kk_ = (S) % 4, ky_ = tap_ / 3, kx_ = tap_ % 3;                                 \
    _Pragma("unroll") for (int mf = 0; mf < MF; ++mf)                                                            \
        wf[SLOT][mf] = *reinterpret_cast<const uint4*>(lds + wb[kk_] + wch + (tap_ * CO + mf * 32) * 128);       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
        xf[SLOT][j] = *reinterpret_cast<const uint4*>(lds + xb[kx_][kk_] + (j + ky_) * kRRowB);                   \
  } while (0)
#define FCVSR_RES_STEP(S)                                                                                        \
  do {                                                                                                           \
    if ((S) + 2 < 36) FCVSR_RES_LOAD(((S) + 2) % 36, ((S) + 2) % 3);                                             \
    _Pragma("unroll") for (int mf = 0; mf < MF; ++mf)                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
            acc[mf][j] = mfma<BF16>(wf[(S) % 3][mf], xf[(S) % 3][j], acc[mf][j]);                                 \
    if ((S) + 2 < 36) __builtin_amdgcn_sched_group_barrier(0x100, MF + 2, 0);                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF, 0);                                                      \
  } while (0)
          FCVSR_RES_LOAD(0, 0);
          FCVSR_RES_LOAD(1, 1);
          __builtin_amdgcn_sched_barrier(0);
          FCVSR_RES_STEP(0);  FCVSR_RES_STEP(1);  FCVSR_RES_STEP(2);  FCVSR_RES_STEP(3);  FCVSR_RES_STEP(4);  FCVSR_RES_STEP(5);
          FCVSR_RES_STEP(6);  FCVSR_RES_STEP(7);  FCVSR_RES_STEP(8);  FCVSR_RES_STEP(9);  FCVSR_RES_STEP(10); FCVSR_RES_STEP(11);
          FCVSR_RES_STEP(12); FCVSR_RES_STEP(13); FCVSR_RES_STEP(14); FCVSR_RES_STEP(15); FCVSR_RES_STEP(16); FCVSR_RES_STEP(17);
          FCVSR_RES_STEP(18); FCVSR_RES_STEP(19); FCVSR_RES_STEP(20); FCVSR_RES_STEP(21); FCVSR_RES_STEP(22); FCVSR_RES_STEP(23);
          FCVSR_RES_STEP(24); FCVSR_RES_STEP(25); FCVSR_RES_STEP(26); FCVSR_RES_STEP(27); FCVSR_RES_STEP(28); FCVSR_RES_STEP(29);
          FCVSR_RES_STEP(30); FCVSR_RES_STEP(31); FCVSR_RES_STEP(32); FCVSR_RES_STEP(33); FCVSR_RES_STEP(34); FCVSR_RES_STEP(35);
#undef FCVSR_RES_STEP
#undef FCVSR_RES_LOAD
          __builtin_amdgcn_s_setprio(0);
        }
        FCVSR_RES_STAMP(1);
        // Bias and activation on the multiplying side, in the accumulator layout (lane = pixel r, registers 4g..4g+3 = couts
        // mf*32 + 8g + 4h + [0,4)): the storing wave of the NEXT phase is the longer of the two roles (it shares its SIMD's issue
        // slots with a partner that has priority), so every VALU instruction moved here shortens the phase.  With a 16-bit
        // destination and no residual the values are also packed here (registers 2g, 2g+1 of each fragment hold quad g).
        if (ch == NCH - 1 && !(a.dbg & 8)) {
          const float* bias_a = reinterpret_cast<const float*>(lds + kRBiasOff) + 4 * h;
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias_a + mf * 32 + 8 * g);
                float v[4] = {acc[mf][j][4 * g] + b4.x, acc[mf][j][4 * g + 1] + b4.y, acc[mf][j][4 * g + 2] + b4.z,
                              acc[mf][j][4 * g + 3] + b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (NSU != 2) v[e] = NSU == 1 ? fmaxf(v[e], ns * v[e]) : fmaxf(v[e], 0.f) + ns * fminf(v[e], 0.f);
                if (FAST) {
                  const uint2 pk = cvt4<BF16>(make_float4(v[0], v[1], v[2], v[3]));
                  acc[mf][j][2 * g] = __uint_as_float(pk.x);
                  acc[mf][j][2 * g + 1] = __uint_as_float(pk.y);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e) acc[mf][j][4 * g + e] = v[e];
                }
              }
        }
      }
      ++k;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS reads are complete before the other phase's copy overwrites
      FCVSR_RES_STAMP(2);
    } else {
      // ================= copy unit k into my buffer; store the tile whose last chunk was unit k - 1 =======================
      ResTile tn = tcur;                             // one decode per tile: the copy below and the stores two phases later share it
      if (k < NU) {
        const int ch = (NCH == 1) ? 0 : (k & (NCH - 1));
        if (ch == 0 && k > 0) tn = res_decode(tabL, a.n_groups, tb + slot + (2 * (k / NCH) + grp) * nslots);
        if (!(a.dbg & 1)) res_stage(tabL, a.zeros, tn, ch, xoff, wq, lane);
      }
      FCVSR_RES_STAMP(3);
      int nst = 0;                                   // store wave-instructions issued below (wave-uniform)
      if (k >= 1 && k <= NU && ((k - 1) & (NCH - 1)) == NCH - 1 && !(a.dbg & 8)) {
        // Branch-free up to the stores: every load of a tile row (bias from LDS, residuals from HBM) is issued before the first
        // use, so a row costs one memory round trip instead of one per 8-channel chunk.
        const ResTile t = tcur;
        const int* T = tabL + t.gi * 32;
        const int GH = T[kTH], GW = T[kTW];
        gchar_t* dbase = tab_ptr(T, kTDstLo);
        const gchar_t* r0base = tab_ptr(T, kTR0Lo);
        const gchar_t* r1base = tab_ptr(T, kTR1Lo);
        const int px = t.tx0 + r;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j) __builtin_amdgcn_sched_barrier(0);    // one row at a time: hoisting both rows' loads spills
          const int py = t.ty0 + 2 * wq + j;
          const bool ok = (py < GH) && (px < GW) && !(a.dbg & 4);
          // lanes outside the image read the residuals of the image's first pixel (always a valid address) and store nothing
          const int pyc = ok ? py : 0, pxc = ok ? px : 0;
          // PixelShuffle(2): rows are packed sub-pixel-major, so this workgroup's 64 couts are 64 consecutive channels of ONE
          // sub-pixel (i, j): the same store at pixel (2y + i, 2x + j), channel n0 % (cout/4)
          const int cq4 = a.cout >> 2, sp = a.ps ? n0 / cq4 : 0, nch = a.ps ? n0 - sp * cq4 : n0;
          const int dyy = a.ps ? 2 * pyc + (sp >> 1) : pyc, dxx = a.ps ? 2 * pxc + (sp & 1) : pxc;
          const unsigned dpix = (unsigned)(t.b * T[kTDstSb] + dyy * T[kTDstSy] + dxx * T[kTDstSx] + nch + 8 * h);   // elements
          const unsigned r0pix = (unsigned)(t.b * T[kTR0Sb] + pyc * T[kTR0Sy] + pxc * T[kTR0Sx] + n0 + 8 * h);
          const unsigned r1pix = (unsigned)(t.b * T[kTR1Sb] + pyc * T[kTR1Sy] + pxc * T[kTR1Sx] + n0 + 8 * h);
          if (FAST) {
            // packed by the multiplying wave: quads 2q, 2q+1 (two dwords each) of the two half-waves -> one 16-byte store
            if (__builtin_amdgcn_ballot_w64(ok) != 0) nst += MF * 2;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
                const u2_t s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mf][j][4 * q]), __float_as_uint(acc[mf][j][4 * q + 2]), false, false);
                const u2_t s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mf][j][4 * q + 1]), __float_as_uint(acc[mf][j][4 * q + 3]), false, false);
                if (ok) gstore(dbase + (size_t)(dpix + mf * 32 + 16 * q) * 2, make_uint4(s0.x, s1.x, s0.y, s1.y));
              }
            continue;
          }
          // accumulator quads 2q, 2q+1 of the two half-waves -> 8 consecutive couts mf*32 + 16q + 8h + [0, 8) of pixel r
          float x[MF][2][8];
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
                const u2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mf][j][8 * q + e]),
                                                                 __float_as_uint(acc[mf][j][8 * q + 4 + e]), false, false);
                x[mf][q][e] = __uint_as_float(sw.x);
                x[mf][q][4 + e] = __uint_as_float(sw.y);
              }
#pragma unroll
          for (int ri = 0; ri < 2; ++ri) {
            if (ri < a.n_res) {
              const gchar_t* rp = ri == 0 ? r0base : r1base;
              const unsigned rpix = ri == 0 ? r0pix : r1pix;
              const float rs = a.rs[ri];
              if (r16) {
                uint4 v[MF][2];
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                  for (int q = 0; q < 2; ++q)
                    v[mf][q] = gload_u4(rp + (size_t)(rpix + mf * 32 + 16 * q) * 2);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                  for (int q = 0; q < 2; ++q) {
                    float rr[8];
                    cvt16x4_to_f32<BF16>(make_uint2(v[mf][q].x, v[mf][q].y), rr);
                    cvt16x4_to_f32<BF16>(make_uint2(v[mf][q].z, v[mf][q].w), rr + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[mf][q][e] = fmaf(rs, rr[e], x[mf][q][e]);
                  }
              } else {
                float4 v[MF][2][2];
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                  for (int q = 0; q < 2; ++q) {
                    v[mf][q][0] = gload_f4(rp + (size_t)(rpix + mf * 32 + 16 * q) * 4);
                    v[mf][q][1] = gload_f4(rp + (size_t)(rpix + mf * 32 + 16 * q + 4) * 4);
                  }
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                  for (int q = 0; q < 2; ++q) {
                    const float4 u0 = v[mf][q][0], u1 = v[mf][q][1];
                    x[mf][q][0] = fmaf(rs, u0.x, x[mf][q][0]); x[mf][q][1] = fmaf(rs, u0.y, x[mf][q][1]);
                    x[mf][q][2] = fmaf(rs, u0.z, x[mf][q][2]); x[mf][q][3] = fmaf(rs, u0.w, x[mf][q][3]);
                    x[mf][q][4] = fmaf(rs, u1.x, x[mf][q][4]); x[mf][q][5] = fmaf(rs, u1.y, x[mf][q][5]);
                    x[mf][q][6] = fmaf(rs, u1.z, x[mf][q][6]); x[mf][q][7] = fmaf(rs, u1.w, x[mf][q][7]);
                  }
              }
            }
          }
          if (__builtin_amdgcn_ballot_w64(ok) != 0) nst += MF * 2 * (DST16 ? 1 : 2);   // the block below runs iff some lane is live
          if (ok) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const float* xx = x[mf][q];
                if (DST16) {
                  const uint2 lo = cvt4<BF16>(make_float4(xx[0], xx[1], xx[2], xx[3])), hi = cvt4<BF16>(make_float4(xx[4], xx[5], xx[6], xx[7]));
                  gstore(dbase + (size_t)(dpix + mf * 32 + 16 * q) * 2, make_uint4(lo.x, lo.y, hi.x, hi.y));
                } else {
                  gstore(dbase + (size_t)(dpix + mf * 32 + 16 * q) * 4, make_float4(xx[0], xx[1], xx[2], xx[3]));
                  gstore(dbase + (size_t)(dpix + mf * 32 + 16 * q + 4) * 4, make_float4(xx[4], xx[5], xx[6], xx[7]));
                }
              }
          }
        }
      }
      tcur = tn;
      FCVSR_RES_STAMP(4);
      // My copies must have landed before the barrier; my stores need not have.  vmcnt counts loads, LDS-DMA and stores
      // together in issue order, and the stores are the youngest operations: leave exactly them outstanding (waiting for them
      // too exposes a full HBM write round trip per phase - measured 78 vs 54 us on a 64->64 layer).
      constexpr int SR = MF * 2 * (DST16 ? 1 : 2);   // stores per tile row
      if (nst == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (nst == SR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SR) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SR) : "memory");
      FCVSR_RES_STAMP(5);
    }
    FCVSR_RES_STAMP(6);
    __builtin_amdgcn_s_barrier();
    if (p >= plast) break;
  }
}


// =====================================================================================================================
// conv3_res3_kernel (round 3): THREE wave groups (768 threads, three waves per SIMD) rotate through three roles, one phase each:
//   copy    : LDS-DMA of the unit's 10 x 34 halo tile into buffer (p + 1) & 1, wait for it (the only role that waits on memory);
//   multiply: the 144 MFMAs per wave of the unit from buffer p & 1 - nothing else, so the SIMD's matrix pipe has a multiplying
//             wave from barrier to barrier;
//   store   : bias + activation + 16-bit packing of the finished tile and its stores.
// In conv3_res_kernel (two groups, two roles) the multiplying role also carried the epilogue arithmetic (1.4 k of its 8.5 k cycles,
// matrix pipe idle) because the other role was full with copy + stores (round-2 stamps; moving the arithmetic
// into the copy / store role, interleaved with the LDS-DMA pieces, measured 10-29 % SLOWER: profiles/NOTES.md).  With the third
// group the phase is the multiply loop alone.  Two halo buffers still suffice: the buffer being filled in phase p is the one read in
// phase p + 1, the one read in phase p is free again in phase p + 1.  16-bit destination without residuals only (the accumulators
// are packed in place); the other destinations stay on conv3_res_kernel.
// The weight rows are staged in a PERMUTED output-channel order (LDS row 8g + 4h + e of a 32-row block holds output channel
// 16h + 4g + e): the 16 accumulator registers of a lane are 16 CONSECUTIVE output channels of its pixel, so the store role needs no
// v_permlane32_swap and a lane's two 16-byte stores are adjacent.  Same products, same summation order: bit-identical results.
template <bool BF16, int NCH, int NSU, int PF>
__global__ __launch_bounds__(768, 3) void conv3_res3_kernel(ResK a) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int CO = 64 / NCH;                       // couts per workgroup
  constexpr int MF = CO / 32;                        // weight (A operand) fragments per wave
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;          // role group 0..2, row pair inside the tile
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

  const int NB = a.cout / CO;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int nb = loc % NB, slot = loc / NB, nslots = (gridDim.x >> 3) / NB;
  const int tb = (int)((long long)xcd * a.total_tiles / 8), te = (int)((long long)(xcd + 1) * a.total_tiles / 8);
  const int n0 = nb * CO;
  const int ntile = (tb + slot < te) ? (te - tb - slot + nslots - 1) / nslots : 0;     // tiles of this workgroup
  if (ntile == 0) return;                                                               // uniform per workgroup
  const int nmine = ntile > grp ? (ntile - grp + 2) / 3 : 0;                            // my tiles: list indices grp, grp + 3, ...
  const int TMAX = (ntile + 2) / 3;                                                     // tiles of group 0 (the most)

  // ---- resident weights, output channels permuted inside every block of 32 rows ----------------------------------------
  {
    const int sub = lane >> 3, cl = lane & 7;
#pragma unroll
    for (int i = 0; i < kRWRows / 8 / 12; ++i) {
      const int g = wave + 12 * i;
      const int row = g * 8 + sub;
      const int q = row / CO, co = row - q * CO;
      const int ch = q / 9, tap = q - ch * 9;
      const int c = cl ^ ((co >> 1) & 7);
      const int co_src = (co & ~31) | (((co >> 2) & 1) << 4) | (((co >> 3) & 3) << 2) | (co & 3);
      const uint16_t* src = a.w + ((long long)tap * a.cout_pad + n0 + co_src) * a.cin_pad + ch * 64 + c * 8;
      glds16(src, __builtin_amdgcn_readfirstlane(lds0 + g * 1024));
    }
  }
  if (tid < CO) reinterpret_cast<float*>(lds + kRBiasOff)[tid] = a.bias ? a.bias[n0 + tid] : 0.f;
  if (tid >= 128 && tid < 128 + 96) reinterpret_cast<int*>(lds + kRTabOff)[tid - 128] = a.tab[(tid - 128) >> 5][(tid - 128) & 31];
  const int* tabL = reinterpret_cast<const int*>(lds + kRTabOff);

  float ns = 1.f;
  if (a.act == FCVSR_ACT_RELU) ns = 0.f;
  else if (a.act == FCVSR_ACT_LEAKY) ns = a.slope;
  else if (a.act == FCVSR_ACT_PRELU) ns = *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(a.slope_ptr));

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // weights, bias and the parameter table are in LDS

  // Group g runs {copy, multiply, store} of its unit in phases g + 3u, g + 3u + 1, g + 3u + 2 (u = unit number): every phase has
  // exactly one copying, one multiplying and one storing group.  The three roles of a unit are straight-line code inside ONE loop
  // iteration, so the accumulators are defined and consumed within an iteration - no loop-carried register tuple for hipcc to
  // shuffle between roles.  Every wave executes the same number of s_barrier instructions: grp + 3 * NCH * TMAX + (2 - grp).
  for (int i = 0; i < grp; ++i) __builtin_amdgcn_s_barrier();
#pragma unroll 1
  for (int ti = 0; ti < TMAX; ++ti) {
    const bool act = ti < nmine;
    // Everything derived from the lane id is recomputed per tile (a few VALU instructions) instead of living in registers across the
    // roles: at three waves per SIMD a wave has 168 registers, and the multiply loop needs 64 accumulators + 48 fragment registers +
    // ~24 LDS address bases.  The empty asm makes the lane id opaque, so hipcc cannot hoist what depends on it.
    int lv = lane, z0 = 0;
    asm volatile("" : "+v"(lv), "+s"(z0));
    const int r = lv & 31, h = lv >> 5;
    const int* tabP = tabL + z0;                     // (same for the parameter table: its LDS reads land in VGPRs)
    const int tile = tb + slot + (3 * ti + grp) * nslots;
    f32x16_t acc[MF][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int p0 = grp + 3 * (ti * NCH + ch);      // phase of this unit's copy; its multiply runs in phase p0 + 1
      const unsigned bsel = ((p0 + 1) & 1) * kRXBytes;
      // ================= copy the unit's halo tile into the buffer the next phase multiplies from =========================
      if (act) {
        if (!(a.dbg & 1)) {
          if (!(a.dbg & 16)) __builtin_amdgcn_s_setprio(1);
          const ResTile t = res_decode(tabP, a.n_groups, tile);
          res_stage(tabP, a.zeros, t, ch, lds0 + kRWBytes + bsel, wq, lv);
          __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      // ================= multiply ==========================================================================================
      if (act) {
        if (ch == 0) {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int i = 0; i < 16; ++i) acc[mf][j][i] = 0.f;
        }
        if (!(a.dbg & 2)) {
          if (a.dbg & 64) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
          const unsigned wch = ch * (9 * CO * 128);
          const unsigned xsel = kRWBytes + bsel;
          unsigned wb[4], xb[3][4];                  // fragment addresses relative to lds
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            wb[kk] = wch + r * 128 + (((2 * kk + h) ^ ((r >> 1) & 7)) << 4);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
              xb[kx][kk] = xsel + (2 * wq * kRHW + r + kx) * 128 + (((2 * kk + h) ^ (((r + kx) >> 1) & 7)) << 4);
          }
          uint4 wf[PF][MF], xf[PF][2];              // fragments are read PF - 1 steps ahead of their MFMAs
#define FCVSR_RES_LOAD(S, SLOT)                                                                                  \
  do {                                                                                                           \
    constexpr int tap_ = (S) / 4, kk_ = (S) % 4, ky_ = tap_ / 3, kx_ = tap_ % 3;                                 \
    _Pragma("unroll") for (int mf = 0; mf < MF; ++mf)                                                            \
        wf[SLOT][mf] = *reinterpret_cast<const uint4*>(lds + wb[kk_] + (tap_ * CO + mf * 32) * 128);             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
        xf[SLOT][j] = *reinterpret_cast<const uint4*>(lds + xb[kx_][kk_] + (j + ky_) * kRRowB);                   \
  } while (0)
#define FCVSR_RES_STEP(S)                                                                                        \
  do {                                                                                                           \
    if ((S) + PF - 1 < 36) FCVSR_RES_LOAD(((S) + PF - 1) % 36, ((S) + PF - 1) % PF);                             \
    _Pragma("unroll") for (int mf = 0; mf < MF; ++mf)                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
            acc[mf][j] = mfma<BF16>(wf[(S) % PF][mf], xf[(S) % PF][j], acc[mf][j]);                               \
    if ((S) + PF - 1 < 36) __builtin_amdgcn_sched_group_barrier(0x100, MF + 2, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF, 0);                                                      \
  } while (0)
          FCVSR_RES_LOAD(0, 0);
          FCVSR_RES_LOAD(1, 1);
          if (PF > 3) FCVSR_RES_LOAD(2, 2);
          __builtin_amdgcn_sched_barrier(0);
          FCVSR_RES_STEP(0);  FCVSR_RES_STEP(1);  FCVSR_RES_STEP(2);  FCVSR_RES_STEP(3);  FCVSR_RES_STEP(4);  FCVSR_RES_STEP(5);
          FCVSR_RES_STEP(6);  FCVSR_RES_STEP(7);  FCVSR_RES_STEP(8);  FCVSR_RES_STEP(9);  FCVSR_RES_STEP(10); FCVSR_RES_STEP(11);
          FCVSR_RES_STEP(12); FCVSR_RES_STEP(13); FCVSR_RES_STEP(14); FCVSR_RES_STEP(15); FCVSR_RES_STEP(16); FCVSR_RES_STEP(17);
          FCVSR_RES_STEP(18); FCVSR_RES_STEP(19); FCVSR_RES_STEP(20); FCVSR_RES_STEP(21); FCVSR_RES_STEP(22); FCVSR_RES_STEP(23);
          FCVSR_RES_STEP(24); FCVSR_RES_STEP(25); FCVSR_RES_STEP(26); FCVSR_RES_STEP(27); FCVSR_RES_STEP(28); FCVSR_RES_STEP(29);
          FCVSR_RES_STEP(30); FCVSR_RES_STEP(31); FCVSR_RES_STEP(32); FCVSR_RES_STEP(33); FCVSR_RES_STEP(34); FCVSR_RES_STEP(35);
#undef FCVSR_RES_STEP
#undef FCVSR_RES_LOAD
          __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS reads are complete before the next phase's copy overwrites
      }
      __builtin_amdgcn_s_barrier();
      // ================= epilogue + stores of the finished tile ===============================================================
      if (act && ch == NCH - 1 && !(a.dbg & 8)) {
        if (a.dbg & 32) __builtin_amdgcn_s_setprio(1);
        const ResTile t = res_decode(tabP, a.n_groups, tile);
        const int* T = tabP + t.gi * 32;
        const int GH = T[kTH], GW = T[kTW];
        gchar_t* dbase = tab_ptr(T, kTDstLo);
        const int px = t.tx0 + r;
        const float* bias_a = reinterpret_cast<const float*>(lds + kRBiasOff) + 16 * h;   // my 16 consecutive couts of each 32-block
        // PixelShuffle(2): rows are packed sub-pixel-major, so this workgroup's 64 couts are 64 consecutive channels of ONE
        // sub-pixel (i, j): the same store at pixel (2y + i, 2x + j), channel n0 % (cout/4)
        const int cq4 = a.cout >> 2, sp = a.ps ? n0 / cq4 : 0, nch = a.ps ? n0 - sp * cq4 : n0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // registers 4g..4g+3 of fragment (mf, j) = couts mf*32 + 16h + 4g + [0, 4) of pixel r: + bias, activation, packed into
          // registers 2g, 2g+1; registers 0..7 then hold the lane's 16 consecutive couts = two adjacent 16-byte stores
          uint2 pk[MF][4];
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 b4 = *reinterpret_cast<const float4*>(bias_a + mf * 32 + 4 * g);
              float v[4] = {acc[mf][j][4 * g] + b4.x, acc[mf][j][4 * g + 1] + b4.y, acc[mf][j][4 * g + 2] + b4.z,
                            acc[mf][j][4 * g + 3] + b4.w};
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (NSU != 2) v[e] = NSU == 1 ? fmaxf(v[e], ns * v[e]) : fmaxf(v[e], 0.f) + ns * fminf(v[e], 0.f);
              pk[mf][g] = cvt4<BF16>(make_float4(v[0], v[1], v[2], v[3]));
            }
          const int py = t.ty0 + 2 * wq + j;
          const bool ok = (py < GH) && (px < GW) && !(a.dbg & 4);
          const int pyc = ok ? py : 0, pxc = ok ? px : 0;
          const int dyy = a.ps ? 2 * pyc + (sp >> 1) : pyc, dxx = a.ps ? 2 * pxc + (sp & 1) : pxc;
          const unsigned dpix = (unsigned)(t.b * T[kTDstSb] + dyy * T[kTDstSy] + dxx * T[kTDstSx] + nch + 16 * h);   // elements
          if (ok) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
              for (int q = 0; q < 2; ++q)
                gstore(dbase + (size_t)(dpix + mf * 32 + 8 * q) * 2, make_uint4(pk[mf][2 * q].x, pk[mf][2 * q].y, pk[mf][2 * q + 1].x, pk[mf][2 * q + 1].y));
          }
        }
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  for (int i = 0; i < 2 - grp; ++i) __builtin_amdgcn_s_barrier();
}

bool conv3_res_supports(int cin, int cout) { return (cin == 64 && cout % 64 == 0) || (cin == 128 && cout % 32 == 0); }

template <bool BF16, int MODE, int NCH, int NSU>
static hipError_t launch_res(const ResArgs& a, hipStream_t st) {
  static DevOnce attr;
  int dev = 0;
  hipError_t e = once_per_device(attr, [&] {
    hipError_t e1 = hipFuncSetAttribute((const void*)conv3_res_kernel<BF16, MODE, NCH, NSU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRLds);
    if constexpr (BF16 && MODE == 2)
      if (e1 == hipSuccess)
        e1 = hipFuncSetAttribute((const void*)conv3_res3_kernel<true, NCH, NSU, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRLds);
    return e1;
  }, &dev);
  if (e != hipSuccess) return e;
  const int cus = device_cu_count(dev);
  if (cus <= 0) return hipErrorInvalidDevice;
  int n_cu = cus > 8 ? cus / 8 * 8 : 8;
  {                                                      // experiment: leave CUs to the other stream's kernels (FCVSR_RES_CUS < CU count)
    static const int lim = getenv("FCVSR_RES_CUS") ? atoi(getenv("FCVSR_RES_CUS")) : 0;
    if (lim >= 8 && lim < n_cu) n_cu = lim / 8 * 8;
  }
  const int NB = a.cout / (64 / NCH);
  // one persistent workgroup per CU; the grid is a multiple of 8 * NB (every XCD gets whole slots of NB cout blocks)
  int grid = n_cu / (8 * NB) * (8 * NB);
  if (grid < 8 * NB) grid = 8 * NB;
  const int need = (a.total_tiles + 7) / 8 * 8 * NB;
  if (grid > need) grid = need;
  ResK k;
  k.n_groups = a.n_groups; k.total_tiles = a.total_tiles; k.cout = a.cout; k.cout_pad = a.cout_pad; k.cin_pad = a.cin_pad;
  k.act = a.act; k.n_res = a.n_res; k.res16 = a.res16; k.dbg = a.dbg; k.ps = a.ps; k.slope = a.slope; k.rs[0] = a.rs[0]; k.rs[1] = a.rs[1];
  k.slope_ptr = a.slope_ptr; k.w = a.w; k.bias = a.bias; k.zeros = a.zeros; k.stamps = a.stamps;
  for (int g = 0; g < 3; ++g) {
    const ResGroup& G = a.g[g < a.n_groups ? g : 0];
    int* T = k.tab[g];
    for (int i = 0; i < 32; ++i) T[i] = 0;
    auto put = [&](int lo, const void* p) { T[lo] = (int)(unsigned)((uintptr_t)p & 0xffffffffu); T[lo + 1] = (int)(unsigned)((uintptr_t)p >> 32); };
    put(kTSrcLo, G.src.p);
    T[kTSrcSb2] = (int)(G.src.sb * 2); T[kTSrcSy2] = (int)(G.src.sy * 2); T[kTSrcSx2] = (int)(G.src.sx * 2);
    T[kTH] = G.H; T[kTW] = G.W; T[kTTilesX] = G.tiles_x; T[kTPerImg] = G.tiles_x * G.tiles_y;
    T[kTTileBegin] = g < a.n_groups ? G.tile_begin : 0x7fffffff;
    put(kTDstLo, G.dst.p); T[kTDstSb] = (int)G.dst.sb; T[kTDstSy] = (int)G.dst.sy; T[kTDstSx] = (int)G.dst.sx;
    const View& r0 = a.n_res > 0 ? G.res[0] : G.dst;      // unused residual slots alias the destination (valid addresses)
    const View& r1 = a.n_res > 1 ? G.res[1] : G.dst;
    put(kTR0Lo, r0.p); T[kTR0Sb] = (int)r0.sb; T[kTR0Sy] = (int)r0.sy; T[kTR0Sx] = (int)r0.sx;
    put(kTR1Lo, r1.p); T[kTR1Sb] = (int)r1.sb; T[kTR1Sy] = (int)r1.sy; T[kTR1Sx] = (int)r1.sx;
  }
  bool v2 = false;
  if constexpr (BF16 && MODE == 2) {     // 16-bit destination without residuals: the three-group kernel
    if (a.variant == 3) {
      hipLaunchKernelGGL((conv3_res3_kernel<true, NCH, NSU, 3>), dim3(grid), dim3(768), kRLds, st, k);
      v2 = true;
    }
  }
  if (!v2) hipLaunchKernelGGL((conv3_res_kernel<BF16, MODE, NCH, NSU>), dim3(grid), dim3(512), kRLds, st, k);
  return hipGetLastError();
}

template <bool BF16, int NCH>
static hipError_t launch_res_mode(const ResArgs& a, bool dst16, hipStream_t st) {
  // act(x) = max(x, ns * x) needs 0 <= ns <= 1: known on the host for every activation but PReLU (slope in device memory)
  const bool nsu = a.act == FCVSR_ACT_NONE || a.act == FCVSR_ACT_RELU || (a.act == FCVSR_ACT_LEAKY && a.slope >= 0.f && a.slope <= 1.f);
  const int mode = !dst16 ? 0 : (a.n_res == 0 ? 2 : 1);
  if (a.act == FCVSR_ACT_NONE) {
    if (mode == 0) return launch_res<BF16, 0, NCH, 2>(a, st);
    if (mode == 1) return launch_res<BF16, 1, NCH, 2>(a, st);
    return launch_res<BF16, 2, NCH, 2>(a, st);
  }
  if (nsu) {
    if (mode == 0) return launch_res<BF16, 0, NCH, 1>(a, st);
    if (mode == 1) return launch_res<BF16, 1, NCH, 1>(a, st);
    return launch_res<BF16, 2, NCH, 1>(a, st);
  }
  if (mode == 0) return launch_res<BF16, 0, NCH, 0>(a, st);
  if (mode == 1) return launch_res<BF16, 1, NCH, 0>(a, st);
  return launch_res<BF16, 2, NCH, 0>(a, st);
}

hipError_t launch_conv3_res(const ResArgs& a, bool bf16, bool dst16, hipStream_t st) {
  if (!conv3_res_supports(a.cin, a.cout)) return hipErrorInvalidValue;
  if (a.cin == 64) return bf16 ? launch_res_mode<true, 1>(a, dst16, st) : launch_res_mode<false, 1>(a, dst16, st);
  return bf16 ? launch_res_mode<true, 2>(a, dst16, st) : launch_res_mode<false, 2>(a, dst16, st);
}

}  // namespace fcvsr
