// gemm_check: dW = dY^T X through pcvae_linear_bwd_weight of a given libpcvae_hip.so against a CPU fp64 result, repeated.
//   g++ -O2 -o build/gemm_check tools/gemm_check.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64 -ldl
//   build/gemm_check <lib.so> [M N K iters]
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <cmath>
#include <ctime>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int (*dw_t)(const float*, int64_t, const float*, int64_t, float*, int64_t, float*, int64_t, int64_t, int64_t, void*);
struct Desc { int32_t kind, act; const float* a; int64_t lda; const float* b; int64_t ldb; float* c; int64_t ldc; const float* aux;
              int64_t ldaux; float* aux_out; int64_t M, N, K; };   // pcvae_gemm_desc
typedef size_t (*wsb_t)(const Desc*, int);
typedef int (*grp_t)(const Desc*, int, void*, size_t, void*);
int main(int argc, char** argv) {
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    dw_t dwf = (dw_t)dlsym(h, "pcvae_linear_bwd_weight");
    wsb_t wsb = (wsb_t)dlsym(h, "pcvae_linear_group_ws_bytes");
    grp_t grp = (grp_t)dlsym(h, "pcvae_linear_group");
    const bool use_ws = wsb && grp && !getenv("CHECK_NO_WS");   // the grouped entry point with its scratch buffer (what ops.py does)
    void* ws = nullptr; const size_t ws_cap = 512u << 20;
    if (use_ws) { CK(hipMalloc(&ws, ws_cap)); CK(hipMemset(ws, 0, ws_cap)); }
    const int64_t M = argc > 2 ? atoll(argv[2]) : 300, N = argc > 3 ? atoll(argv[3]) : 256, K = argc > 4 ? atoll(argv[4]) : 1419;
    const int iters = argc > 5 ? atoi(argv[5]) : 10;
    std::vector<float> hy(M * N), hx(M * K), out(N * K), ob(N);
    uint32_t s = 777;
    for (auto& v : hy) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    std::vector<double> ref((size_t)N * K, 0.0), refb(N, 0.0);
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            const double y = hy[m * N + n];
            refb[n] += y;
            for (int64_t k = 0; k < K; ++k) ref[n * K + k] += y * hx[m * K + k];
        }
    float *Y, *X, *dW, *db;
    CK(hipMalloc(&Y, M * N * 4)); CK(hipMalloc(&X, M * K * 4)); CK(hipMalloc(&dW, N * K * 4)); CK(hipMalloc(&db, N * 4));
    CK(hipMemcpy(Y, hy.data(), M * N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, hx.data(), M * K * 4, hipMemcpyHostToDevice));
    int nbad = 0;
    for (int it = 0; it < iters; ++it) {
        if (getenv("CHECK_H2D_ZERO")) { std::vector<float> z(N * K, 0.f); CK(hipMemcpy(dW, z.data(), N * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, z.data(), N * 4, hipMemcpyHostToDevice)); } else { CK(hipMemset(dW, 0, N * K * 4)); CK(hipMemset(db, 0, N * 4)); }
        CK(hipDeviceSynchronize());
        if (getenv("CHECK_SLEEP_MS")) { struct timespec ts{0, 1000000L * atol(getenv("CHECK_SLEEP_MS"))}; nanosleep(&ts, nullptr); }
        if (getenv("CHECK_DOUBLE_MEMSET")) { CK(hipMemset(db, 0, N * 4)); CK(hipDeviceSynchronize()); }
        if (use_ws) {
            const Desc d{3, 0, Y, N, X, K, dW, K, nullptr, 0, db, M, N, K};
            const size_t need = wsb(&d, 1);
            if (need > ws_cap || grp(&d, 1, ws, need, nullptr) != 0) { printf("group launch failed\n"); return 1; }
        } else if (dwf(Y, N, X, K, dW, K, db, M, N, K, nullptr) != 0) { printf("launch failed\n"); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dW, N * K * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ob.data(), db, N * 4, hipMemcpyDeviceToHost));
        int64_t bad = 0; double worst = 0, wb = 0;
        for (size_t i = 0; i < out.size(); ++i) { const double d = std::fabs(out[i] - ref[i]); if (d > 1e-2) ++bad; if (d > worst) worst = d; }
        for (int64_t n = 0; n < N; ++n) wb = std::fmax(wb, std::fabs(ob[n] - refb[n]));
        if (bad) ++nbad;
        printf("iter %d: %lld bad elements, max err %.4g, bias max err %.4g\n", it, (long long)bad, worst, wb);
    }
    printf("M=%lld N=%lld K=%lld: %d bad of %d\n", (long long)M, (long long)N, (long long)K, nbad, iters);
    return 0;
}
