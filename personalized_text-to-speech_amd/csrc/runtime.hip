// ABI version + per-thread last-error text for libvitsmi.so.
#include <string>
#include "common.h"

namespace {
thread_local std::string g_last_error;
}

namespace vits {
int note_hip_error(hipError_t e, const char* where) {
  g_last_error = std::string(where) + ": " + hipGetErrorString(e);
  return VITS_E_LAUNCH;
}
}  // namespace vits

extern "C" int vits_abi_version(void) { return 16; }
extern "C" const char* vits_last_error(void) { return g_last_error.c_str(); }
