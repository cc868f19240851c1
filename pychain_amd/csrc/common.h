// common.h - error reporting shared by the host-side translation units.
#ifndef PYCHAIN_HIP_COMMON_H_
#define PYCHAIN_HIP_COMMON_H_

#include <cstdarg>
#include <cstdio>

namespace pychain_hip {

char* last_error_buffer();          // thread-local, 512 bytes (api.hip)
extern int g_verbose_level;

// Test / tuning options set through pychain_hip_set_option (api.hip); nullptr = not set.  Nothing on the call
// path reads the environment.
const char* option(const char* name);

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace pychain_hip
#endif
