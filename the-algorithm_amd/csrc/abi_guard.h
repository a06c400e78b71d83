// abi_guard.h -- no C++ exception leaves an extern "C" entry point.
//
// The headers under include/ promise "no function throws or aborts; every function returns a status".  The entry points
// allocate (std::vector, std::string), so they can throw std::bad_alloc / std::length_error -- and an exception that
// reaches an extern "C" frame is std::terminate, i.e. abort() inside the caller's JVM.  Every int-returning entry point
// is therefore a function-try-block that ends in the module's ABI_CATCH:
//
//   int sann_batch_finish(sann_batch_t *b, void *hip_stream) try {
//     ...
//   } ABI_CATCH
//
// with ABI_CATCH defined per module as
//   catch (...) { return abi_guard::caught(fail, <module>_ENOMEM, <module>_EINTERNAL); }
// (`fail` = the module's function that records the thread-local message and returns the code).
#pragma once
#include <exception>
#include <new>
#include <stdexcept>
#include <string>

namespace abi_guard {

// Call from inside a catch (...) block: rethrows the active exception to classify it.
template <class Fail>
int caught(Fail fail, int enomem, int einternal) noexcept {
  try {
    throw;
  } catch (const std::bad_alloc &) {
    try { return fail(enomem, std::string("out of host memory")); } catch (...) { return enomem; }
  } catch (const std::length_error &e) {  // a container sized from a corrupted count
    try { return fail(einternal, std::string("internal error (length_error): ") + e.what()); } catch (...) { return einternal; }
  } catch (const std::exception &e) {
    try { return fail(einternal, std::string("internal error: ") + e.what()); } catch (...) { return einternal; }
  } catch (...) {
    return einternal;
  }
}

}  // namespace abi_guard
