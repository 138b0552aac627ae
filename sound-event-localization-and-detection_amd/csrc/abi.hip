// ABI bookkeeping entry points of libseld_hip.so.
#include "common.h"

namespace seld {
thread_local int g_last_hip_error = 0;
}

extern "C" int seld_abi_version(void) { return 1; }
extern "C" const char* seld_build_arch(void) { return "gfx950"; }
extern "C" int seld_last_hip_error(void) { return seld::g_last_hip_error; }
