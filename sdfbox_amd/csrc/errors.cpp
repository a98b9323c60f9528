// Error channel of the C ABI: int status codes + a thread-local message.
// Replaces the reference's throw-across-FFI (SdfGen/pch.h:20-26).
#include "sdfhip_internal.h"
#include <cstring>

namespace sdfhip {
static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
void clear_error() { g_err[0] = 0; }
}  // namespace sdfhip

extern "C" const char *sdfhip_last_error(void) { return sdfhip::g_err; }
