// Host-side helpers shared by the launchers: error reporting across the C ABI (no exceptions cross it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/i2i_turbo.h"

namespace i2i {
char* error_buffer();                                  // thread-local, defined in capi.hip
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);                    // hipGetLastError() -> status
}  // namespace i2i
