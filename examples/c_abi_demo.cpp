// A host program that uses the hot path through the C ABI alone -- no Python, no torch:
// hipMalloc'd buffers, one HIP stream, the entry points of include/han_hip.h.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude examples/c_abi_demo.cpp \
//         -Lhan_amd -lhan_hip -Wl,-rpath,'$ORIGIN/../han_amd' -o examples/c_abi_demo
//   examples/c_abi_demo N F DEG [bwd] > out.txt
//
// One meta-path, eval mode: X (N,F) -> han_project_fwd -> han_node_attn_fwd over a ring
// graph (row i: itself and the DEG-1 next nodes) -> han_sem_attn_fwd (P = 1).  Inputs come
// from a fixed LCG so that tests/test_gpu_parity.py can rebuild them and check the printed
// numbers against the oracle.  Prints N*64 node-attention outputs, then N*64 embeddings.
// With a 4th argument "bwd" it then runs the training-side entry points on an upstream gradient
// dOut (same LCG): han_node_attn_fwd with the training extras (no dropout) -> han_node_attn_bwd_rows
// -> han_node_attn_bwd_cols over the transposed ring -> han_score_param_bwd -> han_project_bwd, and
// prints dW (F*64), da1, da2 (64 each), db1, db2 (8 each), dc (64).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
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
                           N, F, K, FP, 0.f, 0.f, 0, nullptr, 0, /*keep*/ nullptr, /*flags*/ 0, st));
    HAN_OK(han_node_attn_fwd(drp, dci, nullptr, dH, HAN_DTYPE_F32, nullptr, df1, nullptr, da2, db2, dc, nullptr, dM, D,
                             nullptr, nullptr, nullptr, nullptr, N, N * DEG, K, FP, 0.2f, 0.f, 0.f, 0, nullptr, 0,
                             HAN_ACT_ELU, /*flags*/ 0, /*split*/ nullptr, /*dense*/ nullptr, st));
    HAN_OK(han_sem_attn_fwd(dM, dwo, dbo, duo, dZ, dbeta, N, 1, D, A, /*flags*/ 0, st));
    HIP_OK(hipStreamSynchronize(st));

    std::vector<float> M(N * D), Z(N * D);
    HIP_OK(hipMemcpy(M.data(), dM, M.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(Z.data(), dZ, Z.size() * 4, hipMemcpyDeviceToHost));
    for (float v : M) printf("%.9g\n", v);
    for (float v : Z) printf("%.9g\n", v);
    if (!(argc > 4 && std::string(argv[4]) == "bwd")) return 0;

    // ---- the backward of out = K2(K1(X)) for an upstream gradient dOut, through the C ABI ----
    std::vector<float> dOut(N * D);
    for (auto &v : dOut) v = lcg();
    // transposed ring: source j is a neighbour of the destinations j, j-1, ..., j-DEG+1 (mod N), ascending
    std::vector<int64_t> colptr(N + 1);
    std::vector<int32_t> rowidx((size_t)N * DEG);
    for (int64_t j = 0; j <= N; ++j) colptr[j] = j * DEG;
    for (int64_t j = 0; j < N; ++j) {
        std::vector<int32_t> dst(DEG);
        for (int d = 0; d < DEG; ++d) dst[d] = (int32_t)(((j - d) % N + N) % N);
        std::sort(dst.begin(), dst.end());
        for (int d = 0; d < DEG; ++d) rowidx[j * DEG + d] = dst[d];
    }
    float *ddOut = to_device(dOut);
    int64_t *dcp = to_device(colptr);
    int32_t *dri = to_device(rowidx);
    if (!ddOut || !dcp || !dri) return 2;
    float *dpre, *dlse, *daggp, *dtsum, *ddf1, *ddc, *ddH, *ddf2, *dda1, *dda2, *ddb1, *ddb2, *ddW;
    void *dgs, *ws_rows = nullptr, *ws_sp = nullptr, *ws_pb = nullptr;
    const size_t gs_row = han_gs_row_bytes(K, FP, HAN_DTYPE_F32);
    HIP_OK(hipMalloc(&dpre, N * D * 4 + 16));
    HIP_OK(hipMalloc(&daggp, N * D * 4 + 16));
    HIP_OK(hipMalloc(&dlse, N * K * 4 + 16));
    HIP_OK(hipMalloc(&dtsum, N * K * 4 + 16));
    HIP_OK(hipMalloc(&dgs, N * gs_row + 16));
    HIP_OK(hipMalloc(&ddf1, N * K * 4 + 16));
    HIP_OK(hipMalloc(&ddf2, N * K * 4 + 16));
    HIP_OK(hipMalloc(&ddc, D * 4 + 16));
    HIP_OK(hipMalloc(&ddH, N * D * 4 + 16));
    HIP_OK(hipMalloc(&dda1, K * FP * 4 + 16));
    HIP_OK(hipMalloc(&dda2, K * FP * 4 + 16));
    HIP_OK(hipMalloc(&ddb1, K * 4 + 16));
    HIP_OK(hipMalloc(&ddb2, K * 4 + 16));
    HIP_OK(hipMalloc(&ddW, (size_t)F * D * 4 + 16));
    const size_t wb_rows = han_node_attn_bwd_workspace(N, K, FP), wb_sp = han_score_param_bwd_workspace(N, K, FP),
                 wb_pb = han_project_bwd_workspace(N, F, K, FP);
    if (wb_rows) HIP_OK(hipMalloc(&ws_rows, wb_rows));
    if (wb_sp) HIP_OK(hipMalloc(&ws_sp, wb_sp));
    if (wb_pb) HIP_OK(hipMalloc(&ws_pb, wb_pb));
    HAN_OK(han_node_attn_fwd(drp, dci, nullptr, dH, HAN_DTYPE_F32, nullptr, df1, nullptr, da2, db2, dc, nullptr, dM, D,
                             nullptr, dlse, daggp, dtsum, N, N * DEG, K, FP, 0.2f, 0.f, 0.f, 0, nullptr, 0,
                             HAN_ACT_ELU, 0, nullptr, nullptr, st));
    HAN_OK(han_node_attn_bwd_rows(ddOut, D, dM, D, daggp, dtsum, df1, dlse, dc, nullptr, dgs, HAN_DTYPE_F32, ddf1,
                                  ddc, ws_rows, wb_rows, N, K, FP, HAN_ACT_ELU, st));
    HAN_OK(han_node_attn_bwd_cols(dcp, dri, nullptr, dgs, nullptr, dH, HAN_DTYPE_F32, df2, ddf1, da1, da2, ddH,
                                  ddf2, N, N * DEG, K, FP, 0.2f, 0.f, 0.f, 0, nullptr, 0, 0, 0, /*split*/ nullptr, /*dense*/ nullptr, st));
    HAN_OK(han_score_param_bwd(dH, HAN_DTYPE_F32, ddf1, ddf2, dda1, dda2, ddb1, ddb2, ws_sp, wb_sp, N, K, FP, st));
    HAN_OK(han_project_bwd(dX, HAN_DTYPE_F32, F, ddH, ddW, ws_pb, wb_pb, N, F, K, FP, 0.f, 0, nullptr, 0, /*keep*/ nullptr, st));
    HIP_OK(hipStreamSynchronize(st));
    auto dump = [](const float *d, size_t n) -> int {
        std::vector<float> h(n);
        if (hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        for (float v : h) printf("%.9g\n", v);
        return 0;
    };
    if (dump(ddW, (size_t)F * D) || dump(dda1, K * FP) || dump(dda2, K * FP) || dump(ddb1, K) || dump(ddb2, K) ||
        dump(ddc, D))
        return 2;
    return 0;
}
