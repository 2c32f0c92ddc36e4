// Torch-free reproducer for the "packed-FP32 VALU op reads ds_read_b128 results under a partial lgkmcnt wait" finding
// ("Multi-stream replays").  One file, no dependencies beyond the HIP runtime:
//
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_pk_hazard.hip -o gpurun_out/lds_pk_hazard && gpurun_out/lds_pk_hazard
//
// Victim (stream A): the exact instruction shape of the output phase of rfft_rows_kernel as hipcc's SLP vectoriser emitted
// it - a workgroup writes a known pattern to LDS, barrier, then every lane issues 4 x ds_read_b128 (addresses k and N-k of
// two buffers), waits with `s_waitcnt lgkmcnt(1)` (= reads 1..3 landed: LDS returns in order), and feeds reads 1 and 3 to
//   variant PK     : v_pk_add_f32 (+ the v_pk_add_f32 ... neg_lo/neg_hi subtraction)      [the failing build]
//   variant SCALAR : v_add_f32 / v_sub_f32 on the same registers under the same waits      [the shipped build]
//   (both also with the address VGPR of reads 1-3 placed INSIDE the read's destination range, as in the failing build)
// then `lgkmcnt(0)` and the same for reads 2 and 4.  The sequences are inline asm on fixed physical registers so both
// variants are byte-for-byte the schedule under test, not whatever the compiler picks.  Every result is checked IN the
// kernel against the closed-form expectation (small integers: exact in f32); mismatches are counted per variant, per lane
// and per "which read fed the value".
// Aggressor (stream B): a 256-thread MFMA + LDS loop (4 x ds_read_b128 + 4 x v_mfma_f32_32x32x16_bf16 per iteration out of
// a 64 KiB LDS image) - the load the lean 3x3 convolution puts on a CU's LDS pipeline.
// The program runs each variant solo and under the aggressor and prints one JSON line with the counts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

constexpr int kN = 320;         // transform length of the original failure (row pass, W = 320)
constexpr int kL = 8;           // channel lanes per element  -> Q4 = 2 float4 per element
constexpr int kNL = kN * kL;    // floats per buffer
constexpr int kThreads = 512;

struct Counters {
  unsigned long long bad_ac;    // wrong values computed from reads 1 and 3 (consumed after lgkmcnt(1))
  unsigned long long bad_bd;    // wrong values computed from reads 2 and 4 (consumed after lgkmcnt(0))
  unsigned long long checked;
  unsigned long long lane_hist[64];
};

__device__ __forceinline__ float pat(int buf, int idx, int it) { return (float)(((idx * 7 + it * 13 + buf * 101) & 0x3fff)); }

template <int MODE>
__global__ __launch_bounds__(kThreads) void victim(Counters* cnt, int iters) {
  __shared__ __align__(16) float lds[2 * kNL];
  const int tid = threadIdx.x;
  unsigned long long bad_ac = 0, bad_bd = 0, checked = 0;
  for (int it = 0; it < iters; ++it) {
    const int seed = it + blockIdx.x * 977;
    for (int i = tid; i < 2 * kNL / 4; i += kThreads) {
      const int buf = (i * 4) / kNL, idx = (i * 4) % kNL;
      *reinterpret_cast<float4*>(lds + i * 4) =
          make_float4(pat(buf, idx, seed), pat(buf, idx + 1, seed), pat(buf, idx + 2, seed), pat(buf, idx + 3, seed));
    }
    __syncthreads();
    for (int t = tid; t < (kN / 2 + 1) * (kL / 4); t += kThreads) {
      const int q = t & (kL / 4 - 1), k = t / (kL / 4);
      const int kn = (kN - k) % kN;
      const unsigned a0 = (unsigned)((k * kL + q * 4) * 4), a1 = (unsigned)((kNL + k * kL + q * 4) * 4);
      const unsigned a2 = (unsigned)((kn * kL + q * 4) * 4), a3 = (unsigned)((kNL + kn * kL + q * 4) * 4);
      float s0, s1, s2, s3, d0, d1, d2, d3, t0, t1, t2, t3, e0, e1, e2, e3;
      if (MODE == 1) {
        asm volatile(
            "ds_read_b128 v[40:43], %16\n\t"
            "ds_read_b128 v[44:47], %17\n\t"
            "ds_read_b128 v[48:51], %18\n\t"
            "ds_read_b128 v[52:55], %19\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_pk_add_f32 v[56:57], v[40:41], v[48:49]\n\t"
            "v_pk_add_f32 v[58:59], v[42:43], v[50:51]\n\t"
            "v_pk_add_f32 v[60:61], v[48:49], v[40:41] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 v[62:63], v[50:51], v[42:43] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_pk_add_f32 v[64:65], v[44:45], v[52:53]\n\t"
            "v_pk_add_f32 v[66:67], v[46:47], v[54:55]\n\t"
            "v_pk_add_f32 v[68:69], v[44:45], v[52:53] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 v[70:71], v[46:47], v[54:55] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\t"
            "v_mov_b32 %4, v60\n\tv_mov_b32 %5, v61\n\tv_mov_b32 %6, v62\n\tv_mov_b32 %7, v63\n\t"
            "v_mov_b32 %8, v64\n\tv_mov_b32 %9, v65\n\tv_mov_b32 %10, v66\n\tv_mov_b32 %11, v67\n\t"
            "v_mov_b32 %12, v68\n\tv_mov_b32 %13, v69\n\tv_mov_b32 %14, v70\n\tv_mov_b32 %15, v71\n\t"
            : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(t0), "=&v"(t1),
              "=&v"(t2), "=&v"(t3), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
            : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
              "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",
              "v70", "v71");
      } else if (MODE == 0) {
        asm volatile(
            "ds_read_b128 v[40:43], %16\n\t"
            "ds_read_b128 v[44:47], %17\n\t"
            "ds_read_b128 v[48:51], %18\n\t"
            "ds_read_b128 v[52:55], %19\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_add_f32 v56, v40, v48\n\tv_add_f32 v57, v41, v49\n\tv_add_f32 v58, v42, v50\n\tv_add_f32 v59, v43, v51\n\t"
            "v_sub_f32 v60, v48, v40\n\tv_sub_f32 v61, v49, v41\n\tv_sub_f32 v62, v50, v42\n\tv_sub_f32 v63, v51, v43\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_f32 v64, v44, v52\n\tv_add_f32 v65, v45, v53\n\tv_add_f32 v66, v46, v54\n\tv_add_f32 v67, v47, v55\n\t"
            "v_sub_f32 v68, v44, v52\n\tv_sub_f32 v69, v45, v53\n\tv_sub_f32 v70, v46, v54\n\tv_sub_f32 v71, v47, v55\n\t"
            "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\t"
            "v_mov_b32 %4, v60\n\tv_mov_b32 %5, v61\n\tv_mov_b32 %6, v62\n\tv_mov_b32 %7, v63\n\t"
            "v_mov_b32 %8, v64\n\tv_mov_b32 %9, v65\n\tv_mov_b32 %10, v66\n\tv_mov_b32 %11, v67\n\t"
            "v_mov_b32 %12, v68\n\tv_mov_b32 %13, v69\n\tv_mov_b32 %14, v70\n\tv_mov_b32 %15, v71\n\t"
            : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(t0), "=&v"(t1),
              "=&v"(t2), "=&v"(t3), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
            : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
              "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",
              "v70", "v71");
      } else if (MODE == 3) {
        asm volatile(
            "v_mov_b32 v42, %16\n\tv_mov_b32 v46, %17\n\tv_mov_b32 v49, %18\n\t"
            "ds_read_b128 v[40:43], v42\n\t"
            "ds_read_b128 v[44:47], v46\n\t"
            "ds_read_b128 v[48:51], v49\n\t"
            "ds_read_b128 v[52:55], %19\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_pk_add_f32 v[56:57], v[40:41], v[48:49]\n\t"
            "v_pk_add_f32 v[58:59], v[42:43], v[50:51]\n\t"
            "v_pk_add_f32 v[60:61], v[48:49], v[40:41] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 v[62:63], v[50:51], v[42:43] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_pk_add_f32 v[64:65], v[44:45], v[52:53]\n\t"
            "v_pk_add_f32 v[66:67], v[46:47], v[54:55]\n\t"
            "v_pk_add_f32 v[68:69], v[44:45], v[52:53] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 v[70:71], v[46:47], v[54:55] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\t"
            "v_mov_b32 %4, v60\n\tv_mov_b32 %5, v61\n\tv_mov_b32 %6, v62\n\tv_mov_b32 %7, v63\n\t"
            "v_mov_b32 %8, v64\n\tv_mov_b32 %9, v65\n\tv_mov_b32 %10, v66\n\tv_mov_b32 %11, v67\n\t"
            "v_mov_b32 %12, v68\n\tv_mov_b32 %13, v69\n\tv_mov_b32 %14, v70\n\tv_mov_b32 %15, v71\n\t"
            : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(t0), "=&v"(t1),
              "=&v"(t2), "=&v"(t3), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
            : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
              "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",
              "v70", "v71");
      } else {
        asm volatile(
            "v_mov_b32 v42, %16\n\tv_mov_b32 v46, %17\n\tv_mov_b32 v49, %18\n\t"
            "ds_read_b128 v[40:43], v42\n\t"
            "ds_read_b128 v[44:47], v46\n\t"
            "ds_read_b128 v[48:51], v49\n\t"
            "ds_read_b128 v[52:55], %19\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_add_f32 v56, v40, v48\n\tv_add_f32 v57, v41, v49\n\tv_add_f32 v58, v42, v50\n\tv_add_f32 v59, v43, v51\n\t"
            "v_sub_f32 v60, v48, v40\n\tv_sub_f32 v61, v49, v41\n\tv_sub_f32 v62, v50, v42\n\tv_sub_f32 v63, v51, v43\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_f32 v64, v44, v52\n\tv_add_f32 v65, v45, v53\n\tv_add_f32 v66, v46, v54\n\tv_add_f32 v67, v47, v55\n\t"
            "v_sub_f32 v68, v44, v52\n\tv_sub_f32 v69, v45, v53\n\tv_sub_f32 v70, v46, v54\n\tv_sub_f32 v71, v47, v55\n\t"
            "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\t"
            "v_mov_b32 %4, v60\n\tv_mov_b32 %5, v61\n\tv_mov_b32 %6, v62\n\tv_mov_b32 %7, v63\n\t"
            "v_mov_b32 %8, v64\n\tv_mov_b32 %9, v65\n\tv_mov_b32 %10, v66\n\tv_mov_b32 %11, v67\n\t"
            "v_mov_b32 %12, v68\n\tv_mov_b32 %13, v69\n\tv_mov_b32 %14, v70\n\tv_mov_b32 %15, v71\n\t"
            : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(t0), "=&v"(t1),
              "=&v"(t2), "=&v"(t3), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
            : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
              "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",
              "v70", "v71");
      }
      const float sv[4] = {s0, s1, s2, s3}, dv[4] = {d0, d1, d2, d3}, tv[4] = {t0, t1, t2, t3}, ev[4] = {e0, e1, e2, e3};
      int bad1 = 0, bad2 = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float A = pat(0, k * kL + q * 4 + c, seed), B = pat(1, k * kL + q * 4 + c, seed);
        const float Cc = pat(0, kn * kL + q * 4 + c, seed), D = pat(1, kn * kL + q * 4 + c, seed);
        bad1 += (sv[c] != A + Cc) + (dv[c] != Cc - A);
        bad2 += (tv[c] != B + D) + (ev[c] != B - D);
      }
      bad_ac += bad1;
      bad_bd += bad2;
      checked += 16;
      if (bad1 | bad2) atomicAdd(&cnt->lane_hist[tid & 63], 1ull);
    }
    __syncthreads();
  }
  if (bad_ac) atomicAdd(&cnt->bad_ac, bad_ac);
  if (bad_bd) atomicAdd(&cnt->bad_bd, bad_bd);
  atomicAdd(&cnt->checked, checked);
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ __launch_bounds__(256) void aggressor(float* sink, int iters) {
  extern __shared__ __align__(16) unsigned char alds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 65536 / 16; i += 256) reinterpret_cast<uint4*>(alds)[i] = make_uint4(0x3f803f80u, 0x3f803f80u, i, tid);
  __syncthreads();
  f32x16_t acc0 = {}, acc1 = {};
  const int lane = tid & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o0 = ((lane * 9 + it * 64 + u * 1024) * 16) & 65535, o1 = ((lane * 9 + it * 64 + u * 1024 + 576) * 16) & 65535;
      const uint4 a = *reinterpret_cast<const uint4*>(alds + (o0 & ~15)), b = *reinterpret_cast<const uint4*>(alds + (o1 & ~15));
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b), __builtin_bit_cast(bf16x8_t, a), acc1, 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  if (s == 12345.678f) sink[tid] = s;
}

struct Result {
  unsigned long long bad_ac, bad_bd, checked;
  int bad_launches, launches;
  unsigned long long lane_hist[64];
};

template <int MODE>
static Result run(bool corun, int launches, Counters* d_cnt, float* d_sink, hipStream_t sa, hipStream_t sb) {
  Result r = {};
  r.launches = launches;
  for (int i = 0; i < launches; ++i) {
    CK(hipMemsetAsync(d_cnt, 0, sizeof(Counters), sa));
    CK(hipStreamSynchronize(sa));
    if (corun)
      for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 65536, sb, d_sink, 600);
    hipLaunchKernelGGL(victim<MODE>, dim3(1024), dim3(kThreads), 0, sa, d_cnt, 16);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(sa));
    Counters h;
    CK(hipMemcpy(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost));
    CK(hipStreamSynchronize(sb));
    r.bad_ac += h.bad_ac;
    r.bad_bd += h.bad_bd;
    r.checked += h.checked;
    r.bad_launches += (h.bad_ac | h.bad_bd) != 0;
    for (int l = 0; l < 64; ++l) r.lane_hist[l] += h.lane_hist[l];
  }
  return r;
}

static void print(const char* name, const Result& r, bool last) {
  printf("\"%s\": {\"launches\": %d, \"bad_launches\": %d, \"values_checked\": %llu, \"bad_from_reads_1_3\": %llu, "
         "\"bad_from_reads_2_4\": %llu, \"bad_lanes\": [",
         name, r.launches, r.bad_launches, r.checked, r.bad_ac, r.bad_bd);
  bool first = true;
  for (int l = 0; l < 64; ++l)
    if (r.lane_hist[l]) {
      printf("%s%d", first ? "" : ", ", l);
      first = false;
    }
  printf("]}%s", last ? "" : ", ");
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 300;
  Counters* d_cnt;
  float* d_sink;
  CK(hipMalloc(&d_cnt, sizeof(Counters)));
  CK(hipMalloc(&d_sink, 1024 * sizeof(float)));
  CK(hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  printf("{");
  print("pk_solo", run<1>(false, launches, d_cnt, d_sink, sa, sb), false);
  print("scalar_solo", run<0>(false, launches, d_cnt, d_sink, sa, sb), false);
  print("pk_corun", run<1>(true, launches, d_cnt, d_sink, sa, sb), false);
  print("scalar_corun", run<0>(true, launches, d_cnt, d_sink, sa, sb), false);
  // the compiler's own register assignment in the failing build had the ADDRESS register of reads 1-3 inside each read's
  // destination range (ds_read_b128 v[4:7], v6): the "_addr_in_dst" variants reproduce that too
  print("pk_addr_in_dst_corun", run<3>(true, launches, d_cnt, d_sink, sa, sb), false);
  print("scalar_addr_in_dst_corun", run<2>(true, launches, d_cnt, d_sink, sa, sb), true);
  printf("}\n");
  return 0;
}
