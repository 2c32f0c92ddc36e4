// Shared helpers for libfcvsr_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include "../../include/fcvsr_hip.h"

namespace fcvsr {

void set_error(const char* fmt, ...);

#define FCVSR_CHECK_ARG(cond, msg)                                              \
  do {                                                                          \
    if (!(cond)) {                                                              \
      fcvsr::set_error("%s: %s (%s:%d)", __func__, msg, __FILE__, __LINE__);    \
      return FCVSR_E_ARG;                                                       \
    }                                                                           \
  } while (0)

#define FCVSR_LAUNCH_CHECK()                                                    \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      fcvsr::set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
      return (int)e__;                                                          \
    }                                                                           \
  } while (0)

// device-side copy of fcvsr_view (f32 only kernels use this)
struct View {
  float* p;
  long long sb, sy, sx, sc;
  int c;
};

inline View to_view(const fcvsr_view& v) {
  View o;
  o.p = (float*)v.ptr; o.sb = v.sb; o.sy = v.sy; o.sx = v.sx; o.sc = v.sc; o.c = v.c;
  return o;
}

inline View dense_nhwc(const float* p, int H, int W, int C) {
  View o;
  o.p = (float*)p; o.sb = (long long)H * W * C; o.sy = (long long)W * C; o.sx = C; o.sc = 1; o.c = C;
  return o;
}

inline bool vec4_ok(const fcvsr_view& v) {
  return v.sc == 1 && (v.c % 4 == 0) && (v.sx % 4 == 0) && (v.sy % 4 == 0) && (v.sb % 4 == 0) &&
         (((uintptr_t)v.ptr) % 16 == 0);
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Per-DEVICE run-once guard for per-kernel attributes (hipFuncSetAttribute applies to the current device's copy of the
// function) and other lazily created per-device state.  Thread-safe: the streamed harness and torch's autograd worker thread
// may both enter the library.  One library-wide mutex (core.hip) serialises the first use on a device.
struct DevOnce { unsigned char done[64]; };
std::mutex& device_mutex();
int device_cu_count(int dev);          // multiProcessorCount of `dev`, cached per device; 0 on error
template <class F>
inline hipError_t once_per_device(DevOnce& f, F fn, int* dev_out = nullptr) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (dev_out) *dev_out = dev;
  if (__atomic_load_n(&f.done[dev], __ATOMIC_ACQUIRE)) return hipSuccess;
  std::lock_guard<std::mutex> lock(device_mutex());
  if (!f.done[dev]) {
    e = fn();
    if (e != hipSuccess) return e;
    __atomic_store_n(&f.done[dev], (unsigned char)1, __ATOMIC_RELEASE);
  }
  return hipSuccess;
}

}  // namespace fcvsr
