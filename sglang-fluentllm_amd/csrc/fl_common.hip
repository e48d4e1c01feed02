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

bool fl_mla_use_x() {
  // 128-row-workgroup MLA decode mapping for s_q*H > 64 (mla_decode_fp8_x.hip): default on; FLUENT_MLA_X=0 selects the
  // 64-row mapping (mla_decode_fp8.hip) for every shape.  Read once: the scheduler's part count depends on it.
  static const bool on = [] {
    const char* e = getenv("FLUENT_MLA_X");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}
bool fl_mla_use_y() {
  // role-specialised 64-row MLA decode mapping (mla_decode_fp8_y.hip) for per-token-FP8 KV and s_q*H > 32: default on;
  // FLUENT_MLA_Y=0 falls back to the mappings above.  Read once: the scheduler's part count depends on it.
  static const bool on = [] {
    const char* e = getenv("FLUENT_MLA_Y");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}
int fl_mla_x_rows_per_wg() {
  // query rows one workgroup of mla_decode_fp8_x.hip owns when a request has more than 64: 128 (one workgroup streams a
  // KV part once for all rows; long requests are split along KV) or, with FLUENT_MLA_X_ROWS=64, 64 (row groups of one
  // request run as neighbouring workgroups of one XCD and share the KV stream through its L2; fewer KV splits)
  // With the role-specialised mapping on (fl_mla_use_y) the scheduler counts parts for 64-row workgroups whatever the KV
  // format (get_mla_metadata does not know it): the plain-fp8 format then runs this file's 64-row form too, so that
  // num_parts x row groups still fills the chip.
  static const int rows = [] {
    const char* e = getenv("FLUENT_MLA_X_ROWS");
    return (fl_mla_use_y() || (e != nullptr && atoi(e) == 64)) ? 64 : 128;
  }();
  return rows;
}
extern "C" int fl_version(void) { return 100; }

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
