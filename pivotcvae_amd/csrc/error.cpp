// Host-side error text + ABI version for libpcvae_hip.so.
#include "common.h"
#include <atomic>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

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

int lds_optin(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;   // (kernel, device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return PCVAE_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) on device %d: %s", bytes, dev, hipGetErrorString(e));
        return PCVAE_ELAUNCH;
    }
    done.insert({kernel, dev});
    return PCVAE_OK;
}

// ---- kernel timer ------------------------------------------------------------------------------------------------------------
namespace {
struct TimedLaunch { int tag; hipEvent_t e0, e1; };
std::vector<TimedLaunch>& timed() { static std::vector<TimedLaunch> v; return v; }
std::mutex& timer_mu() { static std::mutex m; return m; }   // the list is process-wide: launches of any thread append under it
std::atomic<bool> g_timer_on{false};
}
bool timer_on() { return g_timer_on.load(std::memory_order_relaxed); }
void timer_events(int tag, hipEvent_t* start, hipEvent_t* stop) {
    TimedLaunch t{tag, nullptr, nullptr};
    hipEventCreate(&t.e0);
    hipEventCreate(&t.e1);
    {
        std::lock_guard<std::mutex> lock(timer_mu());
        timed().push_back(t);
    }
    *start = t.e0;
    *stop = t.e1;
}
}  // namespace pcvae

// enable = 1: start collecting (forgets earlier launches); 0: stop.  The switch and the list are process-wide (every thread's
// instrumented launches are timed while it is on); both are safe to use from several threads.
extern "C" int pcvae_kernel_timer(int enable) {
    using namespace pcvae;
    std::lock_guard<std::mutex> lock(timer_mu());
    for (auto& t : timed()) { hipEventDestroy(t.e0); hipEventDestroy(t.e1); }
    timed().clear();
    g_timer_on.store(enable != 0, std::memory_order_relaxed);
    return PCVAE_OK;
}
// -> number of timed launches so far; fills ms[i] / tags[i] for the first `cap` of them (synchronises on their stop events)
extern "C" int pcvae_kernel_timer_read(float* ms, int* tags, int cap) {
    using namespace pcvae;
    std::lock_guard<std::mutex> lock(timer_mu());
    int n = 0;
    for (auto& t : timed()) {
        if (n < cap && ms && tags) {
            if (hipEventSynchronize(t.e1) != hipSuccess || hipEventElapsedTime(&ms[n], t.e0, t.e1) != hipSuccess) {
                set_error("kernel_timer_read: event %d unreadable", n);
                return PCVAE_ELAUNCH;
            }
            tags[n] = t.tag;
        }
        ++n;
    }
    return n;
}

extern "C" int pcvae_abi_version(void) { return PCVAE_ABI_VERSION; }
extern "C" const char* pcvae_last_error(void) { return pcvae::g_err; }
