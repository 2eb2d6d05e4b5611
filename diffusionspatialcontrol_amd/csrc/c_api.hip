// Version / status entry points of libdsc_hip.so (include/dsc_hip.h).
#include "dsc_hip.h"

extern "C" int dsc_abi_version(void) { return 1; }
extern "C" const char* dsc_target_arch(void) { return "gfx950"; }
extern "C" const char* dsc_status_string(int status) {
    switch (status) {
        case DSC_OK: return "ok";
        case DSC_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size, or inconsistent shapes)";
        case DSC_ERR_UNSUPPORTED: return "unsupported head dim / dtype / alignment for the gfx950 kernels";
        case DSC_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
        case DSC_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown status";
    }
}

// Tuning profile (dsc_set_tuning_profile): read by the launch rules of conv3x3.hip and gemm.hip at launch (= capture) time.
int g_dsc_tuning_profile = DSC_TUNE_LATENCY;
extern "C" int dsc_set_tuning_profile(int profile) {
    if (profile != DSC_TUNE_LATENCY && profile != DSC_TUNE_THROUGHPUT) return DSC_ERR_BAD_ARG;
    g_dsc_tuning_profile = profile;
    return DSC_OK;
}
extern "C" int dsc_get_tuning_profile(void) { return g_dsc_tuning_profile; }
