#include <stdarg.h>

#include <mutex>
#include <set>
#include <utility>

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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that launches on a second GPU has to
// opt that device in as well (the per-process once-flags left every device but the first at the 64 KB default, and the
// return code was dropped).  Remembered per (kernel, device); the device's LDS size is checked; failures surface through the launch
// wrapper's error string.
static thread_local bool g_launch_failed = false;
bool take_launch_failure() {
  const bool f = g_launch_failed;
  g_launch_failed = false;
  return f;
}
static bool ensure_dynamic_lds_impl(const void* kernel, size_t bytes, const char* what);
bool ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what) {
  const bool ok = ensure_dynamic_lds_impl(kernel, bytes, what);
  if (!ok) g_launch_failed = true;
  return ok;
}
static bool ensure_dynamic_lds_impl(const void* kernel, size_t bytes, const char* what) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { hip_fail(e, what); return false; }
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({kernel, dev})) return true;
  int lds = 0;
  e = hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
  if (e == hipSuccess && lds > 0 && bytes > (size_t)160 * 1024) {
    set_error("%s: %zu bytes of dynamic LDS requested, a CDNA4 workgroup has 160 KB", what, bytes);
    return false;
  }
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { hip_fail(e, what); return false; }
  done.insert({kernel, dev});
  return true;
}
}  // namespace pytc

namespace pytc {
// Small registry of integer tuning knobs (kernel variant selection for A/B measurements).
static struct { char key[48]; int value; } g_knobs[64];
static int g_nknobs = 0;
int tuning_get(const char* key, int dflt) {
  for (int i = 0; i < g_nknobs; ++i)
    if (!strcmp(g_knobs[i].key, key)) return g_knobs[i].value;
  return dflt;
}
}  // namespace pytc

extern "C" int pytc_set_tuning(const char* key, int value) {
  if (!key || strlen(key) >= 48) return PYTC_ERR_INVALID;
  for (int i = 0; i < pytc::g_nknobs; ++i)
    if (!strcmp(pytc::g_knobs[i].key, key)) { pytc::g_knobs[i].value = value; return PYTC_OK; }
  if (pytc::g_nknobs >= 64) return PYTC_ERR_INVALID;
  strcpy(pytc::g_knobs[pytc::g_nknobs].key, key);
  pytc::g_knobs[pytc::g_nknobs++].value = value;
  return PYTC_OK;
}

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
