// fatal.hpp — user-facing fatal errors.  The reference prints
// "[racon::...] error: ..." to stderr and exit(1)s (e.g. src/polisher.cpp:64-135,
// src/window.cpp:19-23).  The C++ surface keeps that behaviour; the C ABI used by
// the Python bindings switches to exceptions so a bad input does not kill the
// interpreter.
#pragma once
#include <stdexcept>
#include <string>

namespace racon {

struct FatalError : std::runtime_error { using std::runtime_error::runtime_error; };

void set_fatal_throws(bool on);                 // default: print + exit(1)
[[noreturn]] void fatal(const std::string& message);

}  // namespace racon
