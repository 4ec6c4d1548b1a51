// A host program that uses the hot path through the C ABI alone -- no Python, no torch:
// hipMalloc'd buffers, one HIP stream, the entry points of include/han_hip.h.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude examples/c_abi_demo.cpp \
//         -Lhan_amd -lhan_hip -Wl,-rpath,'$ORIGIN/../han_amd' -o examples/c_abi_demo
//   examples/c_abi_demo N F DEG > out.txt
//
// One meta-path, eval mode: X (N,F) -> han_project_fwd -> han_node_attn_fwd over a ring
// graph (row i: itself and the DEG-1 next nodes) -> han_sem_attn_fwd (P = 1).  Inputs come
// from a fixed LCG so that tests/test_gpu_parity.py can rebuild them and check the printed
// numbers against the oracle.  Prints N*64 node-attention outputs, then N*64 embeddings.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "han_hip.h"

#define HIP_OK(x)                                                                    \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            return 2;                                                                \
        }                                                                            \
    } while (0)
#define HAN_OK(x)                                                                    \
    do {                                                                             \
        int rc_ = (x);                                                               \
        if (rc_ != 0) {                                                              \
            fprintf(stderr, "%s: %s\n", #x, han_error_string(rc_));                  \
            return 3;                                                                \
        }                                                                            \
    } while (0)

static uint32_t lcg_state = 12345u;
static float lcg() {   // uniform in [-0.5, 0.5), 24 bits
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

template <typename T>
static T *to_device(const std::vector<T> &v) {
    T *d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 200;
    const int F = argc > 2 ? atoi(argv[2]) : 24;
    const int DEG = argc > 3 ? atoi(argv[3]) : 5;
    const int K = 8, FP = 8, D = 64, A = 128;
    if (han_abi_version() != HAN_ABI_VERSION) return 1;

    std::vector<float> X(N * F), W((size_t)F * D), a1(K * FP), a2(K * FP), b1(K), b2(K), c(D);
    std::vector<float> wo((size_t)D * A), bo(A), uo(A);
    for (auto &v : X) v = lcg();
    for (auto &v : W) v = lcg() * 0.5f;
    for (auto &v : a1) v = lcg();
    for (auto &v : a2) v = lcg();
    for (auto &v : b1) v = lcg() * 0.2f;
    for (auto &v : b2) v = lcg() * 0.2f;
    for (auto &v : c) v = lcg() * 0.2f;
    for (auto &v : wo) v = lcg() * 0.4f;
    for (auto &v : bo) v = lcg() * 0.2f;
    for (auto &v : uo) v = lcg();
    std::vector<int64_t> rowptr(N + 1);
    std::vector<int32_t> colidx((size_t)N * DEG);
    for (int64_t i = 0; i <= N; ++i) rowptr[i] = i * DEG;
    for (int64_t i = 0; i < N; ++i)
        for (int d = 0; d < DEG; ++d) colidx[i * DEG + d] = (int32_t)((i + d) % N);

    float *dX = to_device(X), *dW = to_device(W), *da1 = to_device(a1), *da2 = to_device(a2);
    float *db1 = to_device(b1), *db2 = to_device(b2), *dc = to_device(c);
    float *dwo = to_device(wo), *dbo = to_device(bo), *duo = to_device(uo);
    int64_t *drp = to_device(rowptr);
    int32_t *dci = to_device(colidx);
    if (!dX || !dW || !da1 || !da2 || !db1 || !db2 || !dc || !dwo || !dbo || !duo || !drp || !dci) return 2;
    float *dH, *df1, *df2, *dM, *dZ, *dbeta;
    void *ws = nullptr;
    HIP_OK(hipMalloc(&dH, N * D * 4 + 16));
    HIP_OK(hipMalloc(&df1, N * K * 4 + 16));
    HIP_OK(hipMalloc(&df2, N * K * 4 + 16));
    HIP_OK(hipMalloc(&dM, N * D * 4 + 16));
    HIP_OK(hipMalloc(&dZ, N * D * 4 + 16));
    HIP_OK(hipMalloc(&dbeta, N * 4 + 16));
    const size_t ws_bytes = han_project_fwd_workspace(N, F, K, FP);
    if (ws_bytes) HIP_OK(hipMalloc(&ws, ws_bytes));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    HAN_OK(han_project_fwd(dX, HAN_DTYPE_F32, F, dW, da1, da2, db1, db2, dH, HAN_DTYPE_F32, df1, df2, ws, ws_bytes,
                           N, F, K, FP, 0.f, 0.f, 0, nullptr, 0, st));
    HAN_OK(han_node_attn_fwd(drp, dci, nullptr, dH, HAN_DTYPE_F32, nullptr, df1, da2, db2, dc, nullptr, dM, D,
                             nullptr, nullptr, nullptr, nullptr, N, N * DEG, K, FP, 0.2f, 0.f, 0.f, 0, nullptr, 0,
                             HAN_ACT_ELU, /*flags*/ 0, nullptr, st));
    HAN_OK(han_sem_attn_fwd(dM, dwo, dbo, duo, dZ, dbeta, N, 1, D, A, st));
    HIP_OK(hipStreamSynchronize(st));

    std::vector<float> M(N * D), Z(N * D);
    HIP_OK(hipMemcpy(M.data(), dM, M.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(Z.data(), dZ, Z.size() * 4, hipMemcpyDeviceToHost));
    for (float v : M) printf("%.9g\n", v);
    for (float v : Z) printf("%.9g\n", v);
    return 0;
}
