// ABI version and error strings of libnerf_amd.
#include "common.h"

extern "C" int nerf_abi_version(void) { return 4; }

extern "C" const char* nerf_error_string(int code) {
    switch (code) {
        case NERF_OK: return "ok";
        case NERF_ERR_ARG: return "invalid argument (null pointer, non-positive size or unsupported option)";
        case NERF_ERR_LAUNCH: return "HIP launch failed";
        case NERF_ERR_SIZE: return "size beyond what the kernel supports";
        case NERF_ERR_RANGE: return "value out of range (e.g. distance beyond the last range bin)";
        default: return "unknown error";
    }
}
