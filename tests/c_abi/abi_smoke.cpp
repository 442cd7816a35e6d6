// C-ABI smoke test without Python or torch: dlopen libvalley_hip.so, run vly_gemm_bf16 (+bias, quick_gelu) and
// vly_rmsnorm on hipMalloc'ed buffers and compare with a host computation.  This is what a non-Python host (the
// C++/cgo/JNI binding of INTEGRATION.md §2) would do.  Built by __graft_entry__.build() with hipcc; run by
// tests/test_kernels_gpu.py::test_c_abi_smoke_binary.   usage: abi_smoke <path to libvalley_hip.so>
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/valley_hip.h"      // VLY_ABI_VERSION (declarations only: every call below goes through dlsym)

#define HIP_OK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP call failed: %s\n", #x); return 2; } } while (0)

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

typedef int (*gemm_fn)(const void*, const void*, const float*, const float*, void*, int, int, int, int, int, int, int, int, int,
                       int, void*);
typedef int (*rms_fn)(const float*, const float*, void*, int, int, float, void*);
typedef int (*ver_fn)(void);
typedef const char* (*err_fn)(void);

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_smoke libvalley_hip.so\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    auto ver = (ver_fn)dlsym(h, "vly_abi_version");
    auto gemm = (gemm_fn)dlsym(h, "vly_gemm_bf16");
    auto rms = (rms_fn)dlsym(h, "vly_rmsnorm");
    auto lasterr = (err_fn)dlsym(h, "vly_last_error");
    if (!ver || !gemm || !rms || !lasterr) { fprintf(stderr, "missing symbol\n"); return 2; }
    if (ver() != VLY_ABI_VERSION) { fprintf(stderr, "ABI version %d\n", ver()); return 2; }

    const int M = 300, N = 264, K = 192;
    std::vector<uint16_t> a(M * K), w(N * K);
    std::vector<float> bias(N);
    for (int i = 0; i < M * K; ++i) a[i] = f2bf(sinf(0.37f * i) * 0.9f);
    for (int i = 0; i < N * K; ++i) w[i] = f2bf(cosf(0.11f * i) * 0.07f);
    for (int i = 0; i < N; ++i) bias[i] = 0.01f * (i % 17) - 0.05f;
    void *da, *dw, *dc;
    float* db;
    HIP_OK(hipMalloc(&da, a.size() * 2));
    HIP_OK(hipMalloc(&dw, w.size() * 2));
    HIP_OK(hipMalloc(&dc, (size_t)M * N * 2));
    HIP_OK(hipMalloc((void**)&db, N * 4));
    HIP_OK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db, bias.data(), N * 4, hipMemcpyHostToDevice));
    int rc = gemm(da, dw, db, nullptr, dc, M, N, K, K, K, N, 0, /*quick_gelu*/ 1, /*bf16*/ 0, /*tile: auto*/ 0, nullptr);
    if (rc) { fprintf(stderr, "vly_gemm_bf16 rc=%d: %s\n", rc, lasterr()); return 1; }
    HIP_OK(hipDeviceSynchronize());
    std::vector<uint16_t> c((size_t)M * N);
    HIP_OK(hipMemcpy(c.data(), dc, c.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float s = bias[n];
            for (int k = 0; k < K; ++k) s += bf2f(a[m * K + k]) * bf2f(w[n * K + k]);
            const float ref = s / (1.f + expf(-1.702f * s));
            const double e = fabs(bf2f(c[(size_t)m * N + n]) - ref) / (fabs(ref) + 0.05);
            if (e > worst) worst = e;
        }
    printf("gemm+bias+quick_gelu: worst relative error %.4g\n", worst);
    if (!(worst < 2e-2)) return 1;

    // error path: K not a multiple of 64 must be rejected with a message, not computed
    rc = gemm(da, dw, db, nullptr, dc, M, N, 100, K, K, N, 0, 0, 0, 0, nullptr);
    if (rc == 0 || !strlen(lasterr())) { fprintf(stderr, "bad K was accepted\n"); return 1; }

    const int R = 7, D = 4096;
    std::vector<float> x((size_t)R * D), g(D);
    for (size_t i = 0; i < x.size(); ++i) x[i] = sinf(0.013f * i) * 3.f;
    for (int i = 0; i < D; ++i) g[i] = 1.f + 0.001f * (i % 13);
    float *dx, *dg;
    void* dy;
    HIP_OK(hipMalloc((void**)&dx, x.size() * 4));
    HIP_OK(hipMalloc((void**)&dg, D * 4));
    HIP_OK(hipMalloc(&dy, x.size() * 2));
    HIP_OK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dg, g.data(), D * 4, hipMemcpyHostToDevice));
    rc = rms(dx, dg, dy, R, D, 1e-5f, nullptr);
    if (rc) { fprintf(stderr, "vly_rmsnorm rc=%d: %s\n", rc, lasterr()); return 1; }
    HIP_OK(hipDeviceSynchronize());
    std::vector<uint16_t> y(x.size());
    HIP_OK(hipMemcpy(y.data(), dy, y.size() * 2, hipMemcpyDeviceToHost));
    worst = 0;
    for (int r = 0; r < R; ++r) {
        double ss = 0;
        for (int i = 0; i < D; ++i) ss += (double)x[(size_t)r * D + i] * x[(size_t)r * D + i];
        const float rstd = 1.f / sqrtf((float)(ss / D) + 1e-5f);
        for (int i = 0; i < D; ++i) {
            const float ref = g[i] * (x[(size_t)r * D + i] * rstd);
            const double e = fabs(bf2f(y[(size_t)r * D + i]) - ref);
            if (e > worst) worst = e;
        }
    }
    printf("rmsnorm: worst abs error %.4g\n", worst);
    if (!(worst < 2e-2)) return 1;
    printf("C ABI smoke: OK\n");
    return 0;
}
