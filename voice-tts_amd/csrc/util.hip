// util.hip -- error text, version.
#include <stdarg.h>

#include "common.h"

namespace ixtts {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ixtts

extern "C" const char* ixtts_version(void) { return "ixtts-hip 0.1.0 (gfx950)"; }
extern "C" const char* ixtts_last_error(void) { return ixtts::g_err; }
