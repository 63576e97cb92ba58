// Stub: under the emulator build <hip/hip_runtime.h> resolves here (tests/emu is first on the
// include path) and hip_emu.h is force-included.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "../hip_emu.h"
