// abi.hip -- library-level entry points of libpdr_hip.so.
#include "pdr_common.h"

namespace {
thread_local char g_last_error[256] = "";
}

namespace pdr {
void set_last_error(hipError_t e) {
  const char* s = hipGetErrorString(e);
  int i = 0;
  for (; s && s[i] && i < 255; ++i) g_last_error[i] = s[i];
  g_last_error[i] = 0;
}
}  // namespace pdr

extern "C" int pdr_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char* pdr_last_error(void) { return g_last_error; }
