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

// ---- the hardware queues of the process that loads this library -------------------------------------------
// The HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues -- 4 unless the environment says otherwise -- and
// streams that share a queue run their kernels one behind the other.  A host that keeps four frames in flight on four streams of
// its own (what the frame pipeline of INTEGRATION.md section 2 does) renders cfg-2's frame in 0.1014 ms on the runtime's four queues
// and in 0.0841 ms on eight: 20 % (tests/c_frames_in_flight.c, a plain C host; profiles/r06_hw_queues_c_host.txt).  The runtime
// reads the variable when it initialises -- at the process's first HIP call, not when libamdhip64.so is mapped (the same file: a
// program that exports it at the top of main() gets the eight queues) -- so the library exports it when IT is loaded, before its
// own kernels are registered, unless the host has set the variable itself (any value: the host's word stands) or SDFHIP_KEEP_ENV
// is set (the library then leaves the environment alone).  A process that initialised HIP before loading the library (a PyTorch
// host) keeps what it had; bench.py exports the variable itself before it imports torch.
__attribute__((constructor(101))) static void sdfhip_export_hw_queues(void)
{
    if (getenv("SDFHIP_KEEP_ENV") == nullptr) (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite*/ 0);
}
