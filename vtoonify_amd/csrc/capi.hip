// C-ABI plumbing shared by every entry point of libvtoonify_amd.so
// (include/vtoonify_amd.h): error text, launch checking, build identification.
#include <stdarg.h>

#include "vt_common.hpp"

static thread_local char g_err[512] = "";

void vt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int vt_check_launch(const char* what) {
#ifdef VT_EMU
    (void)what;
    return VT_OK;
#else
    // hipGetLastError only reports launch-configuration errors; it does not synchronise
    // (the reference op does not check at all: upfirdn2d_kernel.cu:300-360).
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vt_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return VT_ERR_LAUNCH;
    }
    return VT_OK;
#endif
}

extern "C" int vt_abi_version(void) { return VT_ABI_VERSION; }
extern "C" const char* vt_last_error(void) { return g_err; }
extern "C" const char* vt_build_target(void) {
#ifdef VT_EMU
    return "host-emulation";
#else
    return "gfx950";
#endif
}
