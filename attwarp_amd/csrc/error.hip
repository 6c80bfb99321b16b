// Error reporting and version for libattwarp_hip.so.
#include "common.hpp"

#include <atomic>
#include <cstring>

namespace attwarp {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#ifdef ATTWARP_TUNING
// stored as value + 1 so that the zero-initialised table means "automatic" everywhere
static std::atomic<int> g_tune[TUNE_COUNT];
int tune(TuneKey k) { return g_tune[k].load(std::memory_order_relaxed) - 1; }

static const char* const kTuneNames[TUNE_COUNT] = {
    "remap_variant", "remap_rows", "remap_chw_split", "remap_tiled", "remap_tile_ko", "remap_alt", "remap_noswz",
    "remap_ldspad", "remap_nt", "lanczos_variant", "lanczos_rows", "clip_variant", "profiles_variant", "remap_cpw", "remap_skew", "attn_hu", "u8_ahead", "chain_seq", "chain_waves", "remap_cv2_double", "step_prio", "bound", "trace_lo", "trace_hi"};

#endif

}  // namespace attwarp

#ifdef ATTWARP_TUNING
extern "C" int attwarp_debug_set(const char* key, int value, int* previous) {
  using namespace attwarp;
  ATTWARP_REQUIRE(key, "debug_set: null key");
  if (strcmp(key, "reset") == 0) {
    for (auto& v : g_tune) v.store(0, std::memory_order_relaxed);
    return ATTWARP_OK;
  }
  for (int i = 0; i < TUNE_COUNT; ++i)
    if (strcmp(key, kTuneNames[i]) == 0) {
      const int old = g_tune[i].exchange(value < 0 ? 0 : value + 1, std::memory_order_relaxed);
      if (previous) *previous = old - 1;
      return ATTWARP_OK;
    }
  return fail(ATTWARP_E_ARG, "debug_set: unknown key '%s'", key);
}
#endif

extern "C" int attwarp_version(void) { return ATTWARP_VERSION; }
extern "C" const char* attwarp_last_error(void) { return attwarp::error_buffer(); }
