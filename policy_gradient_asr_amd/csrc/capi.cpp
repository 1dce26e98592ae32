// ABI bookkeeping entry points of libpgasr_hip.so.
#include "pgasr_hip.h"

extern "C" int pgasr_abi_version(void) { return PGASR_ABI_VERSION; }

extern "C" const char* pgasr_status_string(int status) {
    switch (status) {
        case PGASR_OK: return "ok";
        case PGASR_ERR_INVALID_ARG: return "invalid argument";
        case PGASR_ERR_LAUNCH: return "HIP launch/runtime error";
        case PGASR_ERR_WORKSPACE: return "workspace missing or too small";
        case PGASR_ERR_UNSUPPORTED: return "size beyond a compiled-in limit";
        case PGASR_ERR_TIMEOUT: return "bounded in-kernel wait timed out";
        default: return "unknown status";
    }
}
