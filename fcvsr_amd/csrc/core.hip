// Error reporting + library identity for libfcvsr_hip.
#include <stdarg.h>
#include <string.h>
#include "common.h"

namespace fcvsr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
std::mutex& device_mutex() {
  static std::mutex mu;
  return mu;
}
int device_cu_count(int dev) {
  static int n_cu[64] = {0};
  if (dev < 0 || dev >= 64) return 0;
  int v = __atomic_load_n(&n_cu[dev], __ATOMIC_ACQUIRE);
  if (v) return v;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  __atomic_store_n(&n_cu[dev], prop.multiProcessorCount, __ATOMIC_RELEASE);
  return prop.multiProcessorCount;
}
}  // namespace fcvsr

extern "C" const char* fcvsr_last_error(void) { return fcvsr::g_err; }
extern "C" int fcvsr_abi_version(void) { return FCVSR_ABI_VERSION; }
extern "C" int fcvsr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
