// fatal.hpp — user-facing fatal errors.  The reference prints
// "[racon::...] error: ..." to stderr and exit(1)s (e.g. src/polisher.cpp:64-135,
// src/window.cpp:19-23).  The C++ surface keeps that behaviour; the C ABI used by
// the Python bindings switches to exceptions so a bad input does not kill the
// interpreter.
#pragma once
#include <exception>
#include <stdexcept>
#include <string>

namespace racon {

struct FatalError : std::runtime_error { using std::runtime_error::runtime_error; };

void set_fatal_throws(bool on);                 // default: print + exit(1)
[[noreturn]] void fatal(const std::string& message);

// Worker threads must never exit(1) the process under the feet of the other threads (static destructors and the HIP
// runtime's teardown would run concurrently with them): while a FatalThrowsScope lives on a thread, fatal() called on
// THAT thread throws FatalError whatever the process-wide mode is.  The thread's owner catches it, and the thread that
// started the work reports it after the join with fatal_from() -- print + exit(1) in CLI mode, rethrow in library mode.
struct FatalThrowsScope { FatalThrowsScope(); ~FatalThrowsScope(); bool previous; };
[[noreturn]] void fatal_from(const std::exception_ptr& error);

}  // namespace racon
