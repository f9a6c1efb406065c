// Error reporting / version entry points of libaptp_hip.so.
#include <stdarg.h>
#include "aptp_common.h"

static thread_local char g_err[512] = "";

void aptp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* aptp_last_error(void) { return g_err; }
extern "C" int aptp_version(void) { return 100; }
