// api.hip — version / error plumbing of the C ABI.
#include "common.h"

namespace lnrf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace lnrf

extern "C" int lnrf_version(void) { return LNRF_VERSION; }
extern "C" const char* lnrf_last_error(void) { return lnrf::g_err; }
