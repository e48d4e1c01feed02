// Error reporting + device queries for libfluent_mi355.
#include "fl_common.h"
#include <mutex>
#include <string>
#include <cstdlib>

static thread_local char g_err[512] = "";

void fl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fl_last_error(void) { return g_err; }

extern "C" int fl_version(void) { return FL_ABI_VERSION; }

extern "C" int fl_device_cu_count(int device, int* cu_count) {
  static int cached[64];
  FL_CHECK_ARG(cu_count != nullptr && device >= 0 && device < 64, "fl_device_cu_count: bad args");
  if (cached[device] == 0) {
    int v = 0;
    hipError_t e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) {
      fl_set_error("fl_device_cu_count: %s", hipGetErrorString(e));
      return FL_ERR_LAUNCH;
    }
    cached[device] = v;
  }
  *cu_count = cached[device];
  return FL_OK;
}
