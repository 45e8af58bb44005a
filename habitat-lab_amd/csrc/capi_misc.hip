// Version / error-string entry points of libhabitat_amd.so.
#include "hab_common.h"
#include "../../include/habitat_amd.h"

extern "C" int hab_abi_version(void) { return HAB_ABI_VERSION; }

extern "C" const char* hab_error_string(int code) {
    switch (code) {
        case HAB_OK: return "ok";
        case HAB_ERR_ARG: return "habitat_amd: invalid argument";
        case HAB_ERR_UNSUPPORTED: return "habitat_amd: unsupported configuration";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "habitat_amd: unknown error";
    }
}
