// Status plumbing of libasr_mi355x: last-error string and version.  Plain C++ (no HIP): also part of the host-only
// AddressSanitizer / UBSan build of the file parsers (make asan).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void asr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* asr_last_error(void) { return g_err; }
extern "C" int asr_version(void) { return 100; }
