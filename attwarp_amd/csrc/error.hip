// Error reporting and version for libattwarp_hip.so.
#include "common.hpp"

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

}  // namespace attwarp

extern "C" int attwarp_version(void) { return ATTWARP_VERSION; }
extern "C" const char* attwarp_last_error(void) { return attwarp::error_buffer(); }
