// Host-side error text + ABI version for libpcvae_hip.so.
#include "common.h"

namespace pcvae {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return PCVAE_ELAUNCH;
    }
    return PCVAE_OK;
}
}  // namespace pcvae

extern "C" int pcvae_abi_version(void) { return PCVAE_ABI_VERSION; }
extern "C" const char* pcvae_last_error(void) { return pcvae::g_err; }
