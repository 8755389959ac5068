// libgmlm_hip.so: error channel, version and device check.
#include <stdarg.h>

#include "common.hpp"

namespace gmlm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gmlm

extern "C" int gmlm_version(void) { return GMLM_ABI_VERSION; }
extern "C" const char* gmlm_last_error(void) { return gmlm::g_err; }

extern "C" int gmlm_device_check(int* cu_count, int* wave_size, char* arch, int arch_len) {
  int dev = 0;
  GMLM_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  GMLM_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    gmlm::set_error("libgmlm_hip.so is built for gfx950 only; current device is %s", prop.gcnArchName);
    return GMLM_EDEVICE;
  }
  return GMLM_OK;
}
