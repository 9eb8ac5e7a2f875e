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
// Measurement only.  Round 6: the switch and the list are PER THREAD (they were process-wide, locked): a thread that turns the timer
// on times its own instrumented launches and nobody else's, reads back its own list, and the library keeps no process-global
// mutable state besides the once-per-(kernel, device) LDS opt-in table above.
namespace {
struct TimedLaunch { int tag; hipEvent_t e0, e1; };
struct ThreadTimer {
    bool on = false;
    std::vector<TimedLaunch> launches;
    void clear() {
        for (auto& t : launches) { hipEventDestroy(t.e0); hipEventDestroy(t.e1); }
        launches.clear();
    }
    ~ThreadTimer() { clear(); }
};
ThreadTimer& thread_timer() { static thread_local ThreadTimer t; return t; }
}
bool timer_on() { return thread_timer().on; }
void timer_events(int tag, hipEvent_t* start, hipEvent_t* stop) {
    TimedLaunch t{tag, nullptr, nullptr};
    hipEventCreate(&t.e0);
    hipEventCreate(&t.e1);
    thread_timer().launches.push_back(t);
    *start = t.e0;
    *stop = t.e1;
}
}  // namespace pcvae

// enable = 1: start collecting the CALLING THREAD's instrumented launches (forgets its earlier ones); 0: stop.
extern "C" int pcvae_kernel_timer(int enable) {
    auto& tt = pcvae::thread_timer();
    tt.clear();
    tt.on = enable != 0;
    return PCVAE_OK;
}
// -> number of launches the calling thread timed so far; fills ms[i] / tags[i] for the first `cap` of them (synchronises on their
// stop events)
extern "C" int pcvae_kernel_timer_read(float* ms, int* tags, int cap) {
    using namespace pcvae;
    int n = 0;
    for (auto& t : thread_timer().launches) {
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
