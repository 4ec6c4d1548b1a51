// K3 -- semantic-level (meta-path) attention, forward and backward (gfx950).
//
// Reference arithmetic: utils/layers.py:152-159 (SimpleAttLayer):
//   v = tanh(M @ Womega + bomega)   (N,P,A)      s = v . uomega   (N,P)
//   beta = softmax over P, PER NODE (the code, not the paper's node average)
//   Z = sum_p beta_p M_p            (N,D)
// v (2 GB at N = 1M, P = 4) is never written: one wave owns one node, holds the
// node's P rows in registers one at a time (lane l = feature l, 256-B coalesced
// loads), keeps Womega in LDS and runs the D x A contraction on the VALU with
// v_readlane broadcasts of the row, then an online softmax over P.
// Roofline: the M / Z streams are HBM-bound (N*(P+1)*256 B); the contraction is
// 2*N*P*D*A flop of fp32 (65.5 GF at N = 1M, P = 4) -- VALU-bound in this version.
#include "han_common.h"

namespace {

__device__ __forceinline__ float bcast_lane(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2/(1+e^{2x}); absolute error ~1e-7, saturates cleanly at +-1
    const float e = __expf(2.f * x);
    return 1.f - __fdividef(2.f, 1.f + e);
}

template <int CA>
__device__ __forceinline__ void row_matvec(const float *Wl, float mval, int lane, const float (&b)[CA],
                                           float (&pre)[CA]) {
    constexpr int A = 64 * CA;
#pragma unroll
    for (int ca = 0; ca < CA; ++ca) pre[ca] = b[ca];
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        const float mk = bcast_lane(mval, k);
        const float *wrow = Wl + k * A + lane * CA;
#pragma unroll
        for (int ca = 0; ca < CA; ++ca) pre[ca] += mk * wrow[ca];
    }
}

template <int CA>
__global__ __launch_bounds__(256) void sem_attn_fwd_kernel(const float *M, const float *Wg, const float *bw,
                                                           const float *uw, float *Z, float *beta, int64_t N,
                                                           int P) {
    constexpr int A = 64 * CA;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;   // [64][A]
    for (int i = threadIdx.x; i < 64 * A; i += 256) Wl[i] = Wg[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float b[CA], u[CA];
#pragma unroll
    for (int ca = 0; ca < CA; ++ca) {
        b[ca] = bw[lane * CA + ca];
        u[ca] = uw[lane * CA + ca];
    }
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t n = wave0; n < N; n += nwaves) {
        float zacc = 0.f, mrun = HAN_NEG_BIG, lrun = 0.f, sreg = 0.f;
        for (int p = 0; p < P; ++p) {
            const float mval = M[(n * P + p) * 64 + lane];
            float pre[CA];
            row_matvec<CA>(Wl, mval, lane, b, pre);
            float part = 0.f;
#pragma unroll
            for (int ca = 0; ca < CA; ++ca) part += fast_tanh(pre[ca]) * u[ca];
            const float s = han_wave_sum(part);
            if (lane == p) sreg = s;
            const float mn = fmaxf(mrun, s);
            const float sc = __expf(mrun - mn), pe = __expf(s - mn);
            lrun = lrun * sc + pe;
            zacc = zacc * sc + pe * mval;
            mrun = mn;
        }
        const float inv = 1.f / lrun;
        Z[n * 64 + lane] = zacc * inv;
        if (lane < P) beta[n * P + lane] = __expf(sreg - mrun) * inv;
    }
}

// slab row per block: [64*A] dW, [A] db, [A] du
template <int CA>
__global__ __launch_bounds__(256) void sem_attn_bwd_kernel(const float *M, const float *Wg, const float *bw,
                                                           const float *uw, const float *beta,
                                                           const float *dZ, float *dM, float *slab,
                                                           int64_t N, int P) {
    constexpr int A = 64 * CA;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;            // [64][A]
    float *WT = smem + 64 * A;   // [A][64]
    for (int i = threadIdx.x; i < 64 * A; i += 256) {
        const float w = Wg[i];
        Wl[i] = w;
        WT[(i % A) * 64 + (i / A)] = w;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float b[CA], u[CA], du[CA], db[CA];
    float dWacc[64][CA];
#pragma unroll
    for (int ca = 0; ca < CA; ++ca) {
        b[ca] = bw[lane * CA + ca];
        u[ca] = uw[lane * CA + ca];
        du[ca] = 0.f;
        db[ca] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 64; ++k)
#pragma unroll
        for (int ca = 0; ca < CA; ++ca) dWacc[k][ca] = 0.f;

    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t n = wave0; n < N; n += nwaves) {
        const float dz = dZ[n * 64 + lane];
        const float breg = lane < P ? beta[n * P + lane] : 0.f;
        float dbreg = 0.f;
        for (int p = 0; p < P; ++p) {
            const float mval = M[(n * P + p) * 64 + lane];
            const float d = han_wave_sum(dz * mval);   // d beta_p = dZ . M_p
            if (lane == p) dbreg = d;
        }
        const float S = han_wave_sum(breg * dbreg);
        const float dsreg = breg * (dbreg - S);        // d s_p on lane p
        for (int p = 0; p < P; ++p) {
            const float mval = M[(n * P + p) * 64 + lane];
            float pre[CA], dpre[CA];
            row_matvec<CA>(Wl, mval, lane, b, pre);
            const float ds = __shfl(dsreg, p, 64);
            const float bp = __shfl(breg, p, 64);
#pragma unroll
            for (int ca = 0; ca < CA; ++ca) {
                const float v = fast_tanh(pre[ca]);
                dpre[ca] = ds * u[ca] * (1.f - v * v);
                du[ca] += ds * v;
                db[ca] += dpre[ca];
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const float mk = bcast_lane(mval, k);
#pragma unroll
                for (int ca = 0; ca < CA; ++ca) dWacc[k][ca] += mk * dpre[ca];
            }
            float dm = bp * dz;
#pragma unroll
            for (int src = 0; src < 64; ++src) {
#pragma unroll
                for (int ca = 0; ca < CA; ++ca) {
                    const float dp = bcast_lane(dpre[ca], src);
                    dm += WT[(src * CA + ca) * 64 + lane] * dp;
                }
            }
            dM[(n * P + p) * 64 + lane] = dm;
        }
    }
    // block reduction of the parameter gradients through LDS, then one slab row
    __syncthreads();
    float *red = smem;   // reuse: [64*A] dW | [A] db | [A] du   (fits: 2*64*A floats available)
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int k = 0; k < 64; ++k)
#pragma unroll
                for (int ca = 0; ca < CA; ++ca) {
                    const int idx = k * A + lane * CA + ca;
                    red[idx] = (w == 0 ? 0.f : red[idx]) + dWacc[k][ca];
                }
#pragma unroll
            for (int ca = 0; ca < CA; ++ca) {
                const int idx = 64 * A + lane * CA + ca;
                red[idx] = (w == 0 ? 0.f : red[idx]) + db[ca];
                red[idx + A] = (w == 0 ? 0.f : red[idx + A]) + du[ca];
            }
        }
        __syncthreads();
    }
    float *out = slab + (int64_t)blockIdx.x * (64 * A + 2 * A);
    for (int i = threadIdx.x; i < 64 * A + 2 * A; i += 256) out[i] = red[i];
}

__global__ void sem_reduce_kernel(const float *slab, int nblocks, int width, int A, float *dW, float *db,
                                  float *du) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= width) return;
    float s = 0.f;
    for (int bk = 0; bk < nblocks; ++bk) s += slab[(int64_t)bk * width + n];
    if (n < 64 * A) dW[n] = s;
    else if (n < 64 * A + A) db[n - 64 * A] = s;
    else du[n - 64 * A - A] = s;
}

constexpr int kSemBwdBlocks = 512;

int sem_grid(int64_t N, int cap) { return han_grid_for(N, 4, cap); }

}  // namespace

extern "C" int han_sem_attn_fwd(const float *M, const float *w_omega, const float *b_omega,
                                const float *u_omega, float *Z, float *beta, int64_t N, int P, int D, int A,
                                void *stream) {
    if (!M || !w_omega || !b_omega || !u_omega || !Z || !beta || N < 0 || P <= 0) return HAN_E_BADARG;
    if (D != HAN_D || P > 64 || (A != 64 && A != 128 && A != 256)) return HAN_E_UNSUPPORTED;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int grid = sem_grid(N, 256 * 4);
    const size_t lds = (size_t)64 * A * sizeof(float);
    hipError_t e = hipSuccess;
    if (A == 64) {
        sem_attn_fwd_kernel<1><<<grid, 256, lds, st>>>(M, w_omega, b_omega, u_omega, Z, beta, N, P);
    } else if (A == 128) {
        sem_attn_fwd_kernel<2><<<grid, 256, lds, st>>>(M, w_omega, b_omega, u_omega, Z, beta, N, P);
    } else {
        e = hipFuncSetAttribute((const void *)sem_attn_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        if (e != hipSuccess) return (int)e;
        sem_attn_fwd_kernel<4><<<grid, 256, lds, st>>>(M, w_omega, b_omega, u_omega, Z, beta, N, P);
    }
    HAN_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t han_sem_attn_bwd_workspace(int64_t N, int P, int D, int A) {
    (void)N; (void)P; (void)D;
    return (size_t)kSemBwdBlocks * (size_t)(64 * A + 2 * A) * sizeof(float);
}

extern "C" int han_sem_attn_bwd(const float *M, const float *w_omega, const float *b_omega,
                                const float *u_omega, const float *beta, const float *dZ, float *dM,
                                float *dw_omega, float *db_omega, float *du_omega, void *workspace,
                                size_t workspace_bytes, int64_t N, int P, int D, int A, void *stream) {
    if (!M || !w_omega || !b_omega || !u_omega || !beta || !dZ || !dM || !dw_omega || !db_omega || !du_omega ||
        !workspace || N < 0 || P <= 0)
        return HAN_E_BADARG;
    if (D != HAN_D || P > 64 || (A != 64 && A != 128)) return HAN_E_UNSUPPORTED;
    if (workspace_bytes < han_sem_attn_bwd_workspace(N, P, D, A)) return HAN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = sem_grid(N > 0 ? N : 1, kSemBwdBlocks);
    const size_t lds = (size_t)2 * 64 * A * sizeof(float);
    float *slab = (float *)workspace;
    if (A == 64) {
        sem_attn_bwd_kernel<1><<<grid, 256, lds, st>>>(M, w_omega, b_omega, u_omega, beta, dZ, dM, slab, N, P);
    } else {
        hipError_t e = hipFuncSetAttribute((const void *)sem_attn_bwd_kernel<2>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        sem_attn_bwd_kernel<2><<<grid, 256, lds, st>>>(M, w_omega, b_omega, u_omega, beta, dZ, dM, slab, N, P);
    }
    HAN_CHECK_LAUNCH();
    const int width = 64 * A + 2 * A;
    sem_reduce_kernel<<<(width + 255) / 256, 256, 0, st>>>(slab, grid, width, A, dw_omega, db_omega, du_omega);
    HAN_CHECK_LAUNCH();
    return 0;
}
