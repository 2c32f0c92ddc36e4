// Small device helpers shared by the MFMA convolution kernels (conv_mfma.hip, conv_res.hip).
#pragma once
#include "common.h"

namespace fcvsr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;


template <bool BF16>
__device__ __forceinline__ uint2 cvt4(float4 v) {
  if (BF16) {
    bf16x4_t c = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    return __builtin_bit_cast(uint2, c);
  } else {
    f16x4_t c = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    return __builtin_bit_cast(uint2, c);
  }
}

template <bool BF16>
__device__ __forceinline__ f32x16_t mfma(uint4 a, uint4 b, f32x16_t c) {
  if (BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <bool BF16>
__device__ __forceinline__ void cvt16x4_to_f32(uint2 v, float* o) {
  if (BF16) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  } else {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 a = __builtin_bit_cast(h2, v.x), b = __builtin_bit_cast(h2, v.y);
    o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)b[0]; o[3] = (float)b[1];
  }
}

// residual load of NV (4 or 8) consecutive channels at element offset `off` of a view base (f32 or 16-bit storage)
template <bool BF16, int NV>
__device__ __forceinline__ void load_res(const float* base, long long off, bool r16, float* o) {
  if (r16) {
    const uint16_t* p = reinterpret_cast<const uint16_t*>(base) + off;
    if (NV == 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(p);
      cvt16x4_to_f32<BF16>(make_uint2(v.x, v.y), o);
      cvt16x4_to_f32<BF16>(make_uint2(v.z, v.w), o + 4);
    } else {
      cvt16x4_to_f32<BF16>(*reinterpret_cast<const uint2*>(p), o);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NV; k += 4) {
      const float4 t = *reinterpret_cast<const float4*>(base + off + k);
      o[k] = t.x; o[k + 1] = t.y; o[k + 2] = t.z; o[k + 3] = t.w;
    }
  }
}

}  // namespace fcvsr
