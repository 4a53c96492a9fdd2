// Error plumbing + device info for libv3d_hip.so.
#include "common.h"

static thread_local char g_err[512] = "";

void v3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int v3d_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        v3d_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return V3D_ERR_LAUNCH;
    }
    return V3D_OK;
}

int v3d_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

extern "C" int v3d_abi_version(void) { return V3D_ABI_VERSION; }
extern "C" const char* v3d_last_error(void) { return g_err; }

extern "C" int v3d_device_info(int32_t* out4) {
    V3D_REQUIRE(out4 != nullptr, "v3d_device_info: null out");
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        v3d_set_error("v3d_device_info: hipGetDevice: %s", hipGetErrorString(e));
        return V3D_ERR_LAUNCH;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        v3d_set_error("v3d_device_info: hipGetDeviceProperties: %s", hipGetErrorString(e));
        return V3D_ERR_LAUNCH;
    }
    out4[0] = prop.multiProcessorCount;
    out4[1] = (int32_t)prop.sharedMemPerBlock;
    out4[2] = prop.warpSize;
    int arch = 0;
    // gcnArchName looks like "gfx950:sramecc+:xnack-"
    if (sscanf(prop.gcnArchName, "gfx%d", &arch) < 1) arch = 0;
    out4[3] = arch;
    return V3D_OK;
}

extern "C" int v3d_sizeof_gemm_args(void) { return (int)sizeof(v3d_gemm_args); }
