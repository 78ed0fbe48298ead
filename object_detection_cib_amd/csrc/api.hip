// Error / version plumbing of the libkodhip C ABI (see include/kodhip.h).
#include "kodhip_common.h"
#include <stdarg.h>

namespace {
thread_local char g_err[512] = "";
}

extern "C" {

void kodhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* kodhip_last_error(void) { return g_err; }

int kodhip_version(void) { return 100; }

// Returns the number of visible HIP devices (0 when there is no GPU); never throws.
int kodhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C"
