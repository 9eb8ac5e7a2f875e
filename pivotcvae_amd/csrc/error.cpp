// Host-side error text + ABI version for libpcvae_hip.so.
#include "common.h"
#include <atomic>
#include <mutex>
#include <cstdint>
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

// Dynamic-LDS opt-in, once per (kernel, device).  It sits on the launch path of every bf16 / split-bf16 / screened catalog kernel, so
// the common case - already done - takes no lock: a small open-addressed table of (kernel -> bitmask of devices) read with two
// relaxed-then-acquire atomic loads.  Only a first call per (kernel, device) takes the mutex, sets the attribute and publishes the bit.
namespace {
constexpr int OPTIN_SLOTS = 128;   // the library has ~40 kernels with dynamic LDS over 64 KB; never fills (checked)
struct OptinSlot { std::atomic<const void*> key{nullptr}; std::atomic<uint64_t> devices{0}; };
OptinSlot g_optin[OPTIN_SLOTS];
inline unsigned optin_hash(const void* k) { return (unsigned)(((uintptr_t)k >> 4) * 2654435761u) % OPTIN_SLOTS; }
}  // namespace

int lds_optin(const void* kernel, int bytes) {
    int dev = 0;
    const hipError_t eg = hipGetDevice(&dev);
    if (eg != hipSuccess) {
        set_error("lds_optin: hipGetDevice: %s", hipGetErrorString(eg));
        return PCVAE_ELAUNCH;
    }
    if (dev < 0 || dev >= 64) {
        set_error("lds_optin: device ordinal %d outside [0, 64)", dev);
        return PCVAE_EINVAL;
    }
    const uint64_t bit = 1ull << dev;
    unsigned h = optin_hash(kernel);
    for (int probe = 0; probe < OPTIN_SLOTS; ++probe, h = (h + 1) % OPTIN_SLOTS) {   // fast path: no lock
        const void* k = g_optin[h].key.load(std::memory_order_acquire);
        if (k == kernel) {
            if (g_optin[h].devices.load(std::memory_order_acquire) & bit) return PCVAE_OK;
            break;
        }
        if (k == nullptr) break;
    }
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    h = optin_hash(kernel);
    int slot = -1;
    for (int probe = 0; probe < OPTIN_SLOTS; ++probe, h = (h + 1) % OPTIN_SLOTS) {
        const void* k = g_optin[h].key.load(std::memory_order_acquire);
        if (k == kernel || k == nullptr) { slot = (int)h; break; }
    }
    if (slot < 0) {
        set_error("lds_optin: table of %d kernels is full", OPTIN_SLOTS);
        return PCVAE_EINVAL;
    }
    if (g_optin[slot].key.load(std::memory_order_acquire) == kernel && (g_optin[slot].devices.load(std::memory_order_acquire) & bit))
        return PCVAE_OK;   // another thread did it while this one waited for the lock
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) on device %d: %s", bytes, dev, hipGetErrorString(e));
        return PCVAE_ELAUNCH;
    }
    g_optin[slot].devices.fetch_or(bit, std::memory_order_release);
    g_optin[slot].key.store(kernel, std::memory_order_release);
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
