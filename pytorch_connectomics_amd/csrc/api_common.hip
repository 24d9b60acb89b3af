#include <stdarg.h>

#include "pytc_common.h"

namespace pytc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
  return PYTC_ERR_HIP;
}
}  // namespace pytc

extern "C" int pytc_abi_version(void) { return PYTC_ABI_VERSION; }
extern "C" const char* pytc_last_error(void) { return pytc::g_err; }
extern "C" int pytc_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) return pytc::hip_fail(e, "device_info");
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return PYTC_OK;
}
