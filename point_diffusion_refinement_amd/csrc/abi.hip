// abi.hip -- library-level entry points of libpdr_hip.so.
#include "pdr_common.h"

#include <atomic>
#include <cstring>

namespace {
thread_local char g_last_error[256] = "";

struct OptRow {
  const char* name;
  int def, lo, hi;
};
// index = pdr::Opt
const OptRow k_opts[pdr::OPT_COUNT] = {
    {"fused_ws", 1, 0, 1},        // 0: uniform-wave layer kernels instead of the wave-specialised ones
    {"narrow_kc32", 1, 0, 1},     // 0: 256-row tiles / 16-channel chunks for outputs of <= 64 channels
    {"fps_wave", 1, 0, 2},        // rank-ordered single-wave FPS: 0 never, 1 up to 256 slots, 2 up to 4096 slots
    {"fps_lean", 1, 0, 1},        // 0: the round-1 resident FPS kernel instead of the instruction-lean one
    {"knn_wave", 1, 0, 1},        // 0: thread-per-query instead of wave-per-query kNN (K <= 8)
    {"gn_fold_small", 1, 0, 1},   // 0: 1024-thread GroupNorm fold workgroups
    {"ws_narrow3", 1, 0, 1},      // 0: two instead of three co-resident workgroups per CU for the 128 x 32 tiles
    {"ws_xcd_order", 1, 0, 2},    // tile order: 0 plain, 1 XCD-local for the gathered layer kernels, 2 for all
    {"deep_chunks", 1, 0, 1},     // 0: the tiny per-point layers on the ordinary tiles / 32-channel chunks
    {"deep_ks", 1, 0, 1},         // 0: those layers without the K split among a workgroup's waves
    {"deep_jobs32", 256, 0, 65536},   // right-sized 32-row-tile launches: at most this many 32 x 128 jobs
    {"deep_jobs64", 512, 0, 65536},   // right-sized 64-row-tile launches: at most this many 64 x 64 jobs
};
std::atomic<int> g_opts[pdr::OPT_COUNT];
std::atomic<bool> g_opts_init{false};

void opts_init() {
  if (!g_opts_init.load(std::memory_order_acquire)) {
    for (int i = 0; i < pdr::OPT_COUNT; ++i) g_opts[i].store(k_opts[i].def, std::memory_order_relaxed);
    g_opts_init.store(true, std::memory_order_release);
  }
}

int opt_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < pdr::OPT_COUNT; ++i)
    if (std::strcmp(name, k_opts[i].name) == 0) return i;
  return -1;
}
}  // namespace

namespace pdr {
void set_last_error(hipError_t e) {
  const char* s = hipGetErrorString(e);
  int i = 0;
  for (; s && s[i] && i < 255; ++i) g_last_error[i] = s[i];
  g_last_error[i] = 0;
}
int option(Opt o) {
  opts_init();
  return g_opts[o].load(std::memory_order_relaxed);
}
}  // namespace pdr

extern "C" int pdr_set_option(const char* name, int value) {
  const int i = opt_index(name);
  if (i < 0 || value < k_opts[i].lo || value > k_opts[i].hi) return PDR_EINVAL;
  opts_init();
  g_opts[i].store(value, std::memory_order_relaxed);
  return PDR_OK;
}

extern "C" int pdr_get_option(const char* name, int* value) {
  const int i = opt_index(name);
  if (i < 0 || !value) return PDR_EINVAL;
  *value = pdr::option(static_cast<pdr::Opt>(i));
  return PDR_OK;
}

extern "C" const char* pdr_option_name(int index) {
  return (index >= 0 && index < pdr::OPT_COUNT) ? k_opts[index].name : nullptr;
}

extern "C" int pdr_version(void) { return 200; /* 0.2.0: see the version history in include/pdr_hip.h */ }
extern "C" const char* pdr_last_error(void) { return g_last_error; }
