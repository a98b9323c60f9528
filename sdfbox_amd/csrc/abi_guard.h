// The exception firewall of the C ABI ("int status codes, never throw/abort across the boundary"; the counter-example
// is the reference's throw across P/Invoke, SdfGen/pch.h:20-26 and dllmain.cpp:112-115, which ends the C# process).
//
// EVERY `extern "C"` definition of the library is a function-try-block that ends in one of the macros below
// (tests/test_abi.py::test_every_entry_point_is_behind_the_exception_firewall reads the sources for it), and so is the body of every
// thread the library starts: a std::bad_alloc from a container, a std::system_error from std::thread / std::mutex, or anything a
// callee throws becomes a status code with sdfhip_last_error() naming the entry point -- never std::terminate in the caller's process.
#pragma once
#include "sdfhip_internal.h"

namespace sdfhip {
// Inside a catch (...) handler: what was thrown -> status code (bad_alloc / length_error: SDFHIP_ERR_NOMEM; a stream's failure:
// SDFHIP_ERR_IO; anything else: SDFHIP_ERR_DEVICE), with the thread's last_error set to "<entry>: <what>".  Allocates nothing.
int abi_caught(const char *entry) noexcept;
}  // namespace sdfhip

#define SDFHIP_ABI_CATCH(entry)            catch (...) { return sdfhip::abi_caught(#entry); }
#define SDFHIP_ABI_CATCH_VOID(entry)       catch (...) { (void)sdfhip::abi_caught(#entry); }
#define SDFHIP_ABI_CATCH_AS(entry, value)  catch (...) { (void)sdfhip::abi_caught(#entry); return (value); }
