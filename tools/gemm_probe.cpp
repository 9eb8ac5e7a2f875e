// gemm_probe: every MLP GEMM of one config-4 train step (forward, input gradient, weight gradient of the nine layers that
// run in ground-truth-pivot training) through the C ABI of a libpcvae_hip.so given on the command line, ITER back-to-back
// launches between one HIP event pair per shape (launch cost amortised: the figure is the kernel's own).
//   g++ -O2 -o build/gemm_probe tools/gemm_probe.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64 -ldl
//   build/gemm_probe pivotcvae_amd/lib/libpcvae_hip.so [M]
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int (*fwd_t)(const float*, int64_t, const float*, int64_t, const float*, float*, int64_t, int64_t, int64_t, int64_t, int, void*);
typedef int (*dx_t)(const float*, int64_t, const float*, int64_t, const float*, int64_t, float*, int64_t, int64_t, int64_t, int64_t, void*);
typedef int (*dw_t)(const float*, int64_t, const float*, int64_t, float*, int64_t, float*, int64_t, int64_t, int64_t, void*);
struct Desc { int32_t kind, act; const float* a; int64_t lda; const float* b; int64_t ldb; float* c; int64_t ldc; const float* aux;
              int64_t ldaux; float* aux_out; int64_t M, N, K; };   // pcvae_gemm_desc
typedef size_t (*wsb_t)(const Desc*, int);
typedef int (*grp_t)(const Desc*, int, void*, size_t, void*);

struct Layer { const char* name; int64_t n_out, k_in; bool need_dx; int64_t dx_cols; };

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: gemm_probe <lib.so> [M] [iters]\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    fwd_t fwd = (fwd_t)dlsym(h, "pcvae_linear_fwd");
    dx_t dxf = (dx_t)dlsym(h, "pcvae_linear_bwd_input");
    dw_t dwf = (dw_t)dlsym(h, "pcvae_linear_bwd_weight");
    wsb_t wsb = (wsb_t)dlsym(h, "pcvae_linear_group_ws_bytes");
    grp_t grp = (grp_t)dlsym(h, "pcvae_linear_group");
    if (!fwd || !dxf || !dwf) { printf("missing symbol\n"); return 1; }
    const bool use_ws = wsb && grp && !getenv("PROBE_NO_WS");   // weight gradients through the scratch buffer (what ops.py does)
    void* ws = nullptr;
    size_t ws_cap = 256u << 20;
    if (use_ws) { CK(hipMalloc(&ws, ws_cap)); CK(hipMemset(ws, 0, ws_cap)); }
    const int64_t M = argc > 2 ? atoll(argv[2]) : 8192;
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    const Layer layers[] = {
        {"prior_1", 128, 139, false, 0}, {"prior_2", 128, 128, true, 128}, {"prior_hd", 32, 128, true, 128},
        {"enc_1", 256, 1419, false, 0},  {"enc_2", 256, 256, true, 256},   {"enc_hd", 32, 256, true, 256},
        {"scm_1", 256, 283, true, 283},  {"scm_1z", 256, 283, true, 16},   {"scm_2", 256, 256, true, 256}, {"scm_3", 1152, 256, true, 256},
    };
    const int64_t maxw = 1419, maxn = 1152;
    float *X, *Y, *W, *dW, *db, *bias, *DX;
    CK(hipMalloc(&X, M * maxw * 4)); CK(hipMalloc(&DX, M * maxw * 4)); CK(hipMalloc(&Y, M * maxn * 4)); CK(hipMalloc(&W, maxn * maxw * 4));
    CK(hipMalloc(&dW, maxn * maxw * 4)); CK(hipMalloc(&db, maxn * 4)); CK(hipMalloc(&bias, maxn * 4));
    std::vector<float> hx(M * maxw);
    uint32_t s = 12345;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    CK(hipMemcpy(X, hx.data(), M * maxw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Y, hx.data(), M * maxn * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hx.data(), maxn * maxw * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dW, 0, maxn * maxw * 4)); CK(hipMemset(db, 0, maxn * 4)); CK(hipMemset(bias, 0, maxn * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot_us = 0, tot_fl = 0, step_us = 0, step_fl = 0;
    auto timeit = [&](const char* kind, const Layer& L, int64_t cols, auto&& call, bool in_step) {
        for (int i = 0; i < 3; ++i) call();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) call();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, fl = 2.0 * M * L.n_out * cols;
        printf("%-4s %-9s M=%5lld N=%4lld K=%4lld cols=%4lld %8.1f us %7.1f TF/s  (%.2f of 157.3)\n", kind, L.name, (long long)M,
               (long long)L.n_out, (long long)L.k_in, (long long)cols, us, fl / us / 1e6, fl / us / 1e6 / 157.3);
        tot_us += us; tot_fl += fl;
        if (in_step) { step_us += us; step_fl += fl; }
    };
    for (const Layer& L : layers) {
        const bool z_only = L.dx_cols == 16;
        if (!z_only) {
            timeit("fwd", L, L.k_in, [&] { fwd(X, L.k_in, W, L.k_in, bias, Y, L.n_out, M, L.n_out, L.k_in, 1, nullptr); }, true);
            timeit("dW", L, L.k_in, [&] {
                if (use_ws) {
                    const Desc d{3, 0, Y, L.n_out, X, L.k_in, dW, L.k_in, nullptr, 0, db, M, L.n_out, L.k_in};
                    const size_t need = wsb(&d, 1);
                    if (need > ws_cap || grp(&d, 1, ws, need, nullptr) != 0) { printf("group launch failed\n"); exit(1); }
                } else dwf(Y, L.n_out, X, L.k_in, dW, L.k_in, db, M, L.n_out, L.k_in, nullptr);
            }, true);
        }
        if (L.need_dx)   // dX[M, cols] = dY[M, n_out] . W[n_out, :cols]; X doubles as the activated input (mask source)
            timeit("dX", L, L.dx_cols, [&] { dxf(Y, L.n_out, W, L.k_in, X, L.k_in, DX, L.k_in, M, L.n_out, L.dx_cols, nullptr); },
                   getenv("PROBE_ZCOLS") ? (z_only || L.k_in != 283) : !z_only);
    }
    printf("step GEMMs: %.1f us, %.2f GFLOP, %.1f TF/s = %.3f of the f32 MFMA peak\n", step_us, step_fl / 1e9, step_fl / step_us / 1e6,
           step_fl / step_us / 1e6 / 157.3);
    return 0;
}
