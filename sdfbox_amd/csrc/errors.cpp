// Error channel of the C ABI: int status codes + a thread-local message.
// Replaces the reference's throw-across-FFI (SdfGen/pch.h:20-26).
#include "abi_guard.h"
#include <cstring>
#include <exception>
#include <ios>
#include <new>
#include <stdexcept>
#include <system_error>

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

// The translation behind SDFHIP_ABI_CATCH*: called inside a catch (...) handler, rethrows to learn the type.  fail() formats
// into a thread-local array, so nothing here allocates (the usual reason to be here is that allocation has just failed).
int abi_caught(const char *entry) noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(SDFHIP_ERR_NOMEM, "%s: out of host memory (std::bad_alloc)", entry);
    } catch (const std::length_error &e) {
        return fail(SDFHIP_ERR_NOMEM, "%s: a container would exceed its maximum size (%s)", entry, e.what());
    } catch (const std::ios_base::failure &e) {           // (a std::system_error since C++11: before it)
        return fail(SDFHIP_ERR_IO, "%s: stream error (%s)", entry, e.what());
    } catch (const std::system_error &e) {
        return fail(SDFHIP_ERR_DEVICE, "%s: the system refused a thread or a lock (std::system_error %d: %s)", entry, e.code().value(), e.what());
    } catch (const std::exception &e) {
        return fail(SDFHIP_ERR_DEVICE, "%s: unexpected C++ exception: %s", entry, e.what());
    } catch (...) {
        return fail(SDFHIP_ERR_DEVICE, "%s: unexpected exception of unknown type", entry);
    }
}
}  // namespace sdfhip

extern "C" const char *sdfhip_last_error(void) { return sdfhip::g_err; }     // (reads a thread-local array: nothing to guard)
