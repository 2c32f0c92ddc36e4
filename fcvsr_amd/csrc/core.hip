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
}  // namespace fcvsr

extern "C" const char* fcvsr_last_error(void) { return fcvsr::g_err; }
extern "C" int fcvsr_abi_version(void) { return FCVSR_ABI_VERSION; }
extern "C" int fcvsr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
