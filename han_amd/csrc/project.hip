// K1 -- dense feature projection for all K heads of one meta-path (gfx950).
//
// Reference arithmetic: utils/layers.py:18-24 (per head: input dropout, 1x1
// conv1d == X @ W_k, two 1x1 conv1d == H_k . a + b) and :31-32 (dropout of the
// projected rows, applied after the scores were taken from the undropped rows).
//
// The only GEMM-shaped work on the path, so the only MFMA user: exact-fp32
// v_mfma_f32_16x16x4_f32 (no xf32/TF32 on gfx950).  M = N rows, N = D = 64
// columns, K = F.  A block of 4 waves owns 64 rows x all 64 columns, so X is
// read from HBM exactly once; W (F x 64) is re-read per block from L2.
// Roofline at F = 256: 2*N*F*64 flop vs N*(F+64)*4 B -> ~24 flop/B, i.e. at the
// fp32-MFMA ridge: HBM-bound in eval, MFMA-bound in training where the per-head
// input dropout (layers.py:18-19 sits inside the per-head call) forces one
// masked MFMA per head per 16-column tile.
#include "han_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64;      // rows per block (forward)
constexpr int BK = 32;      // K-step
constexpr int XS_LD = 34;   // LDS leading dims chosen conflict-free for the MFMA fragment reads
constexpr int WS_LD = 80;

struct ProjFwdArgs {
    const void *X;      // fp32 or bf16 (x_bf16), row stride ldx ELEMENTS
    int64_t ldx;
    const float *W;
    void *H;            // fp32 or bf16 (h_bf16), 64 elements per row
    int x_bf16, h_bf16;
    int64_t N;
    int F;
    uint32_t seed_lo, seed_hi, thr_in, thr_fts;   // thr_fts < 2^16: stamp keep bits into H
    const uint64_t *seed_dev;
    float inv_keep_in;
    int64_t row_offset;
    // split-F for short inputs (few row blocks, long reduction): blockIdx.y owns the
    // features [y*f_chunk, (y+1)*f_chunk) and writes a raw partial tile to `partial`
    // ([nsplit][N][64] fp32); project_finish_kernel sums them in a fixed order.
    int f_chunk;        // multiple of BK; >= F when not split
    float *partial;     // null when not split
};

__device__ __forceinline__ float load_x1(const void *X, int bf, int64_t idx) {
    if (bf) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(X)[idx] << 16);
    return reinterpret_cast<const float *>(X)[idx];
}
__device__ __forceinline__ float4_t load_x4(const void *X, int bf, int64_t idx) {   // idx % 4 == 0, aligned
    if (bf) {
        const uint2 w = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(X) + idx);
        float4_t v;
        v[0] = __uint_as_float(w.x << 16);
        v[1] = __uint_as_float(w.x & 0xFFFF0000u);
        v[2] = __uint_as_float(w.y << 16);
        v[3] = __uint_as_float(w.y & 0xFFFF0000u);
        return v;
    }
    return *reinterpret_cast<const float4_t *>(reinterpret_cast<const float *>(X) + idx);
}

// heads covered by one 16-column MFMA tile
template <int FP>
struct HeadsPerTile { static constexpr int value = FP >= 16 ? 1 : 16 / FP; };

// MT = 16-row MFMA tiles per wave (block = 64*MT rows); VEC = 16-byte X loads
// (needs F % 4 == 0, ldx % 4 == 0 and a 16-byte aligned X).  With dropout every
// head a tile covers gets its own accumulator: the A fragment is masked per head
// and the columns of the other heads are simply not read back.
template <int FP, bool DROP, int MT, bool VEC>
__global__ __launch_bounds__(256) void project_fwd_kernel(const ProjFwdArgs a_in) {
    ProjFwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int K = HAN_D / FP;
    constexpr int KQ = (K + 3) / 4;   // one RNG call = four 16-bit draws = four heads
    constexpr int HPT = DROP ? HeadsPerTile<FP>::value : 1;
    constexpr int BMR = 64 * MT;      // rows per block
    constexpr int XL = (BMR * BK) / 256;   // X elements per thread per tile
    __shared__ float Xs[BMR * XS_LD];
    __shared__ float Ws[BK * WS_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * BMR;
    const int k_begin = (int)blockIdx.y * a.f_chunk;
    const int k_end = (k_begin + a.f_chunk < a.F) ? k_begin + a.f_chunk : a.F;

    f32x4 acc[MT][4][HPT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int hh = 0; hh < HPT; ++hh) acc[m][t][hh] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float xr[XL];
    float4_t wr4[2];
    auto load_tile = [&](int k0) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < XL / 4; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 3, c4 = (idx & 7) * 4;
                const int64_t row = row0 + r;
                float4_t v = {0.f, 0.f, 0.f, 0.f};
                if (row < a.N && k0 + c4 < k_end) v = load_x4(a.X, a.x_bf16, row * a.ldx + k0 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[4 * i + e] = v[e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 5, cc = idx & 31;
                const int64_t row = row0 + r;
                xr[i] = (row < a.N && k0 + cc < k_end) ? load_x1(a.X, a.x_bf16, row * a.ldx + k0 + cc) : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int kw = k0 + (idx >> 4);
            wr4[i] = kw < k_end ? *reinterpret_cast<const float4_t *>(a.W + (int64_t)kw * HAN_D + (idx & 15) * 4)
                              : (float4_t){0.f, 0.f, 0.f, 0.f};
        }
    };
    load_tile(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();   // previous tile's fragment reads are done
        if (VEC) {
#pragma unroll
            for (int i = 0; i < XL / 4; ++i) {
                const int idx = tid + 256 * i;
                float *dst = Xs + (idx >> 3) * XS_LD + (idx & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = xr[4 * i + e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const int idx = tid + 256 * i;
                Xs[(idx >> 5) * XS_LD + (idx & 31)] = xr[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4_t *>(Ws + (idx >> 4) * WS_LD + (idx & 15) * 4) = wr4[i];
        }
        __syncthreads();
        if (k0 + BK < k_end) load_tile(k0 + BK);   // in flight under the MFMAs
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            float bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[t] = Ws[(kk + l4) * WS_LD + 16 * t + l15];
            const uint32_t kglob = (uint32_t)(k0 + kk + l4);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int lr = 16 * (w * MT + m) + l15;
                const float av = Xs[lr * XS_LD + kk + l4];
                const uint32_t nglob = (uint32_t)(row0 + lr + a.row_offset);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!DROP) {
                        acc[m][t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[m][t][0], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int hh = 0; hh < HPT; ++hh) {
                            const int head = (16 * t) / FP + hh;
                            // one call serves four heads (identical calls are CSE'd)
                            const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_SEQ, nglob,
                                                            kglob * (uint32_t)KQ + (uint32_t)(head >> 2));
                            const float am = rn.field(head & 3) < a.thr_in ? av : 0.f;
                            acc[m][t][hh] = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bv[t], acc[m][t][hh], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int myhh = HPT > 1 ? l15 / FP : 0;   // which head accumulator holds this lane's column
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 16 * (w * MT + m) + l4 * 4 + r;
                float v = acc[m][t][0][r];
#pragma unroll
                for (int hh = 1; hh < HPT; ++hh) v = (myhh == hh) ? acc[m][t][hh][r] : v;
                if (DROP) v *= a.inv_keep_in;
                if (a.partial) {   // split-F: raw partial sums, finished by project_finish_kernel
                    if (row < a.N) a.partial[((int64_t)blockIdx.y * a.N + row) * HAN_D + 16 * t + l15] = v;
                    continue;
                }
                uint32_t keepbit = 0, stamp = 0;
                if (a.thr_fts < HAN_KEEP_ALL) {
                    // projected-row dropout (layers.py:31-32): the keep bit rides in the lowest
                    // mantissa bit of the STORED element (fp32 bit 0 / bf16 bit 0)
                    const int d = 16 * t + l15;
                    const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_FTS,
                                                    (uint32_t)(row + a.row_offset), (uint32_t)(d >> 2));
                    keepbit = rn.field(d & 3) < a.thr_fts ? 1u : 0u;
                    stamp = 1;
                }
                if (row < a.N) {
                    if (a.h_bf16) {
                        uint32_t b = han_f32_to_bf16_bits(v);
                        if (stamp) b = (b & ~1u) | keepbit;
                        reinterpret_cast<uint16_t *>(a.H)[row * HAN_D + 16 * t + l15] = (uint16_t)b;
                    } else {
                        if (stamp) v = __uint_as_float((__float_as_uint(v) & ~1u) | keepbit);
                        reinterpret_cast<float *>(a.H)[row * HAN_D + 16 * t + l15] = v;
                    }
                }
            }
        }
    }
}

// split-F epilogue: H[row] = sum over the f-chunks (fixed order), then the same keep-bit
// stamping / bf16 rounding as the unsplit kernel's epilogue.
template <bool BF>
__global__ __launch_bounds__(256) void project_finish_kernel(const ProjFwdArgs a_in, int nsplit) {
    ProjFwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    const int q = threadIdx.x & 15;
    const int64_t grp0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t ngrp = (int64_t)gridDim.x * 16;
    for (int64_t row = grp0; row < a.N; row += ngrp) {
        float4_t v = {0.f, 0.f, 0.f, 0.f};
        for (int sidx = 0; sidx < nsplit; ++sidx) {
            const float4_t pv = *reinterpret_cast<const float4_t *>(a.partial + ((int64_t)sidx * a.N + row) * HAN_D + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += pv[e];
        }
        if (a.thr_fts < HAN_KEEP_ALL) {   // layers.py:31-32, d = 4q + e -> counter d/4 = q, field e
            const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_FTS, (uint32_t)(row + a.row_offset),
                                            (uint32_t)q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t keepbit = rn.field(e) < a.thr_fts ? 1u : 0u;
                if (BF) {
                    const uint32_t b = (han_f32_to_bf16_bits(v[e]) & ~1u) | keepbit;
                    v[e] = __uint_as_float(b << 16);
                } else {
                    v[e] = __uint_as_float((__float_as_uint(v[e]) & ~1u) | keepbit);
                }
            }
        }
        han_store_row4<BF>(a.H, row, q, v);
    }
}

// Row-local epilogue: f1 = H_k.a1 + b1, f2 = H_k.a2 + b2 (layers.py:23-24), taken from
// the rows exactly as stored (i.e. including the keep bits stamped in training).
struct ScoreArgs {
    const void *H;
    const float *a1, *a2, *b1, *b2;
    float *f1, *f2;
    int64_t N;
};

template <int FP, bool BF>
__global__ __launch_bounds__(256) void project_scores_kernel(const ScoreArgs a) {
    constexpr int K = HAN_D / FP;
    const int q = threadIdx.x & 15;
    const int head = (4 * q) / FP;
    const int64_t grp0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t ngrp = (int64_t)gridDim.x * 16;
    const float4_t a14 = *reinterpret_cast<const float4_t *>(a.a1 + 4 * q);
    const float4_t a24 = *reinterpret_cast<const float4_t *>(a.a2 + 4 * q);
    const float b1 = a.b1[head], b2 = a.b2[head];
    for (int64_t row = grp0; row < a.N; row += ngrp) {
        const float4_t h4 = han_load_row4<BF>(a.H, row, q);
        float s1 = h4[0] * a14[0] + h4[1] * a14[1] + h4[2] * a14[2] + h4[3] * a14[3];
        float s2 = h4[0] * a24[0] + h4[1] * a24[1] + h4[2] * a24[2] + h4[3] * a24[3];
#pragma unroll
        for (int o = 1; o < FP / 4; o <<= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        if ((4 * q) % FP == 0) {
            a.f1[row * K + head] = s1 + b1;
            a.f2[row * K + head] = s2 + b2;
        }
    }
}

// ---------------------------------------------------------------------------
// backward: dW = X~^T dH.  Output tile 64 (f) x 64 (d) per block, reduction over
// a chunk of rows; partial tiles go to a slab, a second kernel sums the chunks
// (deterministic, no float atomics).
// ---------------------------------------------------------------------------
constexpr int BN = 32;       // reduction step (rows of X / dH)
constexpr int TS_LD = 80;

struct ProjBwdArgs {
    const void *X;
    int x_bf16;
    int64_t ldx;
    const float *dH;
    float *slab;   // [nchunks][F][64]
    int64_t N;
    int F;
    int64_t rows_per_chunk;
    uint32_t seed_lo, seed_hi, thr_in;
    const uint64_t *seed_dev;
    float inv_keep_in;
    int64_t row_offset;
};

// MT = 16-row (f) MFMA tiles per wave: the block owns 64*MT rows of dW, so dH is
// re-read F/(64*MT) times; VEC = 16-byte fp32 X loads.  With dropout every head a
// column tile covers has its own accumulator (A masked per head), as in the forward.
template <int FP, bool DROP, int MT, bool VEC>
__global__ __launch_bounds__(256) void project_bwd_kernel(const ProjBwdArgs a_in) {
    ProjBwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int K = HAN_D / FP;
    constexpr int KQ = (K + 3) / 4;
    constexpr int HPT = DROP ? HeadsPerTile<FP>::value : 1;
    constexpr int BFR = 64 * MT;              // f rows of dW per block
    constexpr int XLD = BFR + 16;             // (XLD mod 32) == 16: conflict-free fragment reads
    constexpr int XL = (BN * BFR) / 256;      // X elements per thread per step
    __shared__ float Xs[BN * XLD];
    __shared__ float Gs[BN * TS_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int f0 = blockIdx.x * BFR;
    const int64_t chunk = blockIdx.y;
    const int64_t n_begin = chunk * a.rows_per_chunk;
    const int64_t n_end = (n_begin + a.rows_per_chunk < a.N) ? n_begin + a.rows_per_chunk : a.N;

    f32x4 acc[MT][4][HPT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int hh = 0; hh < HPT; ++hh) acc[m][t][hh] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float xr[XL];
    float4_t gr4[2];
    auto load_tile = [&](int64_t n0) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < XL / 4; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx / (BFR / 4), c4 = (idx % (BFR / 4)) * 4;
                const int64_t row = n0 + r;
                float4_t v = {0.f, 0.f, 0.f, 0.f};
                if (row < n_end && f0 + c4 < a.F) v = load_x4(a.X, a.x_bf16, row * a.ldx + f0 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[4 * i + e] = v[e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx / BFR, cc = idx % BFR;
                const int64_t row = n0 + r;
                xr[i] = (row < n_end && f0 + cc < a.F) ? load_x1(a.X, a.x_bf16, row * a.ldx + f0 + cc) : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int64_t row = n0 + (idx >> 4);
            gr4[i] = row < n_end ? *reinterpret_cast<const float4_t *>(a.dH + row * HAN_D + (idx & 15) * 4)
                                 : (float4_t){0.f, 0.f, 0.f, 0.f};
        }
    };
    load_tile(n_begin);
    for (int64_t n0 = n_begin; n0 < n_end; n0 += BN) {
        __syncthreads();
        if (VEC) {
#pragma unroll
            for (int i = 0; i < XL / 4; ++i) {
                const int idx = tid + 256 * i;
                *reinterpret_cast<float4_t *>(Xs + (idx / (BFR / 4)) * XLD + (idx % (BFR / 4)) * 4) =
                    (float4_t){xr[4 * i], xr[4 * i + 1], xr[4 * i + 2], xr[4 * i + 3]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const int idx = tid + 256 * i;
                Xs[(idx / BFR) * XLD + (idx % BFR)] = xr[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4_t *>(Gs + (idx >> 4) * TS_LD + (idx & 15) * 4) = gr4[i];
        }
        __syncthreads();
        if (n0 + BN < n_end) load_tile(n0 + BN);
#pragma unroll
        for (int kk = 0; kk < BN; kk += 4) {
            float bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[t] = Gs[(kk + l4) * TS_LD + 16 * t + l15];
            const uint32_t nglob = (uint32_t)(n0 + kk + l4 + a.row_offset);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                // A[i = f][k = n] = X[n][f]
                const int lf = 16 * (w * MT + m) + l15;
                const float av = Xs[(kk + l4) * XLD + lf];
                const uint32_t fglob = (uint32_t)(f0 + lf);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!DROP) {
                        acc[m][t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[m][t][0], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int hh = 0; hh < HPT; ++hh) {
                            const int head = (16 * t) / FP + hh;
                            const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_SEQ, nglob,
                                                            fglob * (uint32_t)KQ + (uint32_t)(head >> 2));
                            const float am = rn.field(head & 3) < a.thr_in ? av : 0.f;
                            acc[m][t][hh] = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bv[t], acc[m][t][hh], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    float *out = a.slab + chunk * (int64_t)a.F * HAN_D;
    const int myhh = HPT > 1 ? l15 / FP : 0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = f0 + 16 * (w * MT + m) + l4 * 4 + r;
                float v = acc[m][t][0][r];
#pragma unroll
                for (int hh = 1; hh < HPT; ++hh) v = (myhh == hh) ? acc[m][t][hh][r] : v;
                if (f < a.F) out[(int64_t)f * HAN_D + 16 * t + l15] = DROP ? v * a.inv_keep_in : v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// backward w.r.t. the INPUT (layers >= 1 of a multi-layer stack, models/gat.py:48-57):
//   dX[n,f] = sum_k  m_k[n,f]/keep * sum_f'  dH[n, k*FP+f'] * W[f, k*FP+f']
// One wave per 16 rows; per head a (16 x FP).(FP x 16) MFMA product, masked by that
// head's input-dropout draw and accumulated.  dX rows may be strided (a slice of the
// previous layer's dM).
// ---------------------------------------------------------------------------
struct ProjBwdInArgs {
    const float *dH, *W;
    float *dX;
    int64_t ldo;
    int64_t N;
    int F;
    uint32_t seed_lo, seed_hi, thr_in;
    const uint64_t *seed_dev;
    float inv_keep_in;
    int64_t row_offset;
};

template <int FP, bool DROP>
__global__ __launch_bounds__(256) void project_bwd_input_kernel(const ProjBwdInArgs a_in) {
    ProjBwdInArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int K = HAN_D / FP;
    constexpr int KQ = (K + 3) / 4;
    constexpr int KS = (FP + 3) / 4;          // MFMA k-steps per head (FP = 4 -> 1, 8 -> 2, ...)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t ntiles = (a.N + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t r0 = tile * 16;
        const int64_t ra = r0 + l15 < a.N ? r0 + l15 : a.N - 1;
        // A fragments: dH[row = l15][k*FP + 4s + l4]
        float af[K][KS];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                const int col = 4 * s2 + l4;
                af[k][s2] = col < FP ? a.dH[ra * HAN_D + k * FP + col] : 0.f;
            }
        for (int f0 = 0; f0 < a.F; f0 += 16) {
            const int f = f0 + l15;
            const int fc = f < a.F ? f : a.F - 1;
            f32x4 out = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                f32x4 acc = DROP ? (f32x4){0.f, 0.f, 0.f, 0.f} : out;
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2) {
                    const int col = 4 * s2 + l4;
                    const float b = col < FP ? a.W[(int64_t)fc * HAN_D + k * FP + col] : 0.f;   // B[kk][j=f]
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k][s2], b, acc, 0, 0, 0);
                }
                if (DROP) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const uint32_t nglob = (uint32_t)(r0 + 4 * l4 + reg + a.row_offset);
                        const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_SEQ, nglob,
                                                        (uint32_t)f * (uint32_t)KQ + (uint32_t)(k >> 2));
                        out[reg] += rn.field(k & 3) < a.thr_in ? acc[reg] : 0.f;
                    }
                } else {
                    out = acc;
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = r0 + 4 * l4 + reg;
                if (row < a.N && f < a.F) a.dX[row * a.ldo + f] = DROP ? out[reg] * a.inv_keep_in : out[reg];
            }
        }
    }
}

bool fp_ok(int K, int FP) {
    return K * FP == HAN_D && (FP == 4 || FP == 8 || FP == 16 || FP == 32 || FP == 64);
}

constexpr int kBwdMT = 2;   // project_bwd: 128 f rows per block

void bwd_geometry(int64_t N, int F, int *ftiles, int64_t *rows_per_chunk, int64_t *nchunks) {
    *ftiles = (F + 64 * kBwdMT - 1) / (64 * kBwdMT);
    int64_t target = 1024 / *ftiles;
    if (target < 1) target = 1;
    int64_t rpc = (N + target - 1) / target;
    rpc = ((rpc + BN - 1) / BN) * BN;
    if (rpc < BN) rpc = BN;
    *rows_per_chunk = rpc;
    *nchunks = N > 0 ? (N + rpc - 1) / rpc : 1;
}

}  // namespace

#define HAN_DISPATCH_FP(FPV, ...)                                 \
    switch (FPV) {                                                \
        case 4: { constexpr int FPC = 4; __VA_ARGS__; } break;    \
        case 8: { constexpr int FPC = 8; __VA_ARGS__; } break;    \
        case 16: { constexpr int FPC = 16; __VA_ARGS__; } break;  \
        case 32: { constexpr int FPC = 32; __VA_ARGS__; } break;  \
        default: { constexpr int FPC = 64; __VA_ARGS__; } break;  \
    }

// Forward geometry.  Long inputs: 128-row blocks, one block per row tile.  Short inputs
// (fewer row tiles than CUs, e.g. ACM: N = 3025, F = 1870): 64-row blocks and the
// reduction over F split into chunks so that ~2 blocks per CU are in flight.
static void fwd_geometry(int64_t N, int F, int *mt, int *nsplit, int *f_chunk) {
    *mt = 2; *nsplit = 1; *f_chunk = ((F + BK - 1) / BK) * BK;
    const int64_t tiles128 = (N + 127) / 128;
    if (tiles128 >= 256 || F < 4 * BK) return;
    *mt = 1;
    const int64_t tiles64 = (N + 63) / 64;
    int64_t want = (512 + tiles64 - 1) / tiles64;
    const int64_t max_split = (F + 2 * BK - 1) / (2 * BK);      // at least two K-steps per chunk
    if (want > max_split) want = max_split;
    if (want <= 1) return;
    const int chunk = (int)(((F + want - 1) / want + BK - 1) / BK) * BK;
    *f_chunk = chunk;
    *nsplit = (F + chunk - 1) / chunk;
}

extern "C" size_t han_project_fwd_workspace(int64_t N, int F, int K, int FP) {
    (void)K; (void)FP;
    int mt, nsplit, f_chunk;
    fwd_geometry(N > 0 ? N : 0, F, &mt, &nsplit, &f_chunk);
    return nsplit > 1 ? (size_t)nsplit * (size_t)N * HAN_D * sizeof(float) : 0;
}

extern "C" int han_project_fwd(const void *X, int x_dtype, int64_t ldx, const float *W, const float *a1,
                               const float *a2, const float *b1, const float *b2, void *H, int table_dtype,
                               float *f1, float *f2, void *workspace, size_t workspace_bytes, int64_t N, int F,
                               int K, int FP, float in_drop, float fts_drop, uint64_t seed,
                               const uint64_t *seed_dev, int64_t row_offset, void *stream) {
    if (N == 0) return 0;   // nothing to do; row pointers of empty tensors may be null
    if (!X || !W || !a1 || !a2 || !b1 || !b2 || !H || !f1 || !f2 || N < 0 || F <= 0 || ldx < F)
        return HAN_E_BADARG;
    if (!fp_ok(K, FP)) return HAN_E_UNSUPPORTED;
    if ((x_dtype != HAN_DTYPE_F32 && x_dtype != HAN_DTYPE_BF16) ||
        (table_dtype != HAN_DTYPE_F32 && !(table_dtype == HAN_DTYPE_BF16 && FP == 8)))
        return HAN_E_UNSUPPORTED;
    if (in_drop < 0.f || in_drop >= 1.f || fts_drop < 0.f || fts_drop >= 1.f) return HAN_E_BADARG;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    ProjFwdArgs a;
    a.X = X; a.ldx = ldx; a.W = W; a.H = H; a.N = N; a.F = F;
    a.x_bf16 = x_dtype == HAN_DTYPE_BF16; a.h_bf16 = table_dtype == HAN_DTYPE_BF16;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_dev = seed_dev;
    a.thr_in = in_drop > 0.f ? han_keep_threshold(1.f - in_drop) : HAN_KEEP_ALL;
    a.thr_fts = fts_drop > 0.f ? han_keep_threshold(1.f - fts_drop) : HAN_KEEP_ALL;
    a.inv_keep_in = 1.f / (1.f - in_drop);
    a.row_offset = row_offset;
    const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)X & (a.x_bf16 ? 7 : 15)) == 0);
    int mt, nsplit;
    fwd_geometry(N, F, &mt, &nsplit, &a.f_chunk);
    a.partial = nullptr;
    if (nsplit > 1) {
        if (!workspace || workspace_bytes < han_project_fwd_workspace(N, F, K, FP)) return HAN_E_WORKSPACE;
        a.partial = (float *)workspace;
    }
    const dim3 grid((unsigned)((N + 64 * mt - 1) / (64 * mt)), (unsigned)nsplit);
#define HAN_LAUNCH_FWD(MTC)                                                                  \
    HAN_DISPATCH_FP(FP, {                                                                    \
        if (in_drop > 0.f) {                                                                 \
            if (vec) project_fwd_kernel<FPC, true, MTC, true><<<grid, 256, 0, st>>>(a);      \
            else project_fwd_kernel<FPC, true, MTC, false><<<grid, 256, 0, st>>>(a);         \
        } else {                                                                             \
            if (vec) project_fwd_kernel<FPC, false, MTC, true><<<grid, 256, 0, st>>>(a);     \
            else project_fwd_kernel<FPC, false, MTC, false><<<grid, 256, 0, st>>>(a);        \
        }                                                                                    \
    })
    if (mt == 2) { HAN_LAUNCH_FWD(2) } else { HAN_LAUNCH_FWD(1) }
#undef HAN_LAUNCH_FWD
    HAN_CHECK_LAUNCH();
    if (nsplit > 1) {
        const int fgrid = han_grid_for(N, 16, 256 * 8);
        if (a.h_bf16) project_finish_kernel<true><<<fgrid, 256, 0, st>>>(a, nsplit);
        else project_finish_kernel<false><<<fgrid, 256, 0, st>>>(a, nsplit);
        HAN_CHECK_LAUNCH();
    }
    ScoreArgs s;
    s.H = H; s.a1 = a1; s.a2 = a2; s.b1 = b1; s.b2 = b2; s.f1 = f1; s.f2 = f2;
    s.N = N;
    const int sgrid = han_grid_for(N, 16, 256 * 8);
    if (a.h_bf16) {
        project_scores_kernel<8, true><<<sgrid, 256, 0, st>>>(s);
    } else {
        HAN_DISPATCH_FP(FP, { project_scores_kernel<FPC, false><<<sgrid, 256, 0, st>>>(s); })
    }
    HAN_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t han_project_bwd_workspace(int64_t N, int F, int K, int FP) {
    (void)K; (void)FP;
    int ftiles; int64_t rpc, nch;
    bwd_geometry(N, F, &ftiles, &rpc, &nch);
    return (size_t)nch * (size_t)F * HAN_D * sizeof(float);
}

extern "C" int han_project_bwd(const void *X, int x_dtype, int64_t ldx, const float *dH, float *dW, void *workspace,
                               size_t workspace_bytes, int64_t N, int F, int K, int FP, float in_drop,
                               uint64_t seed, const uint64_t *seed_dev, int64_t row_offset, void *stream) {
    if (!X || !dH || !dW || !workspace || N < 0 || F <= 0 || ldx < F) return HAN_E_BADARG;
    if (!fp_ok(K, FP)) return HAN_E_UNSUPPORTED;
    if (x_dtype != HAN_DTYPE_F32 && x_dtype != HAN_DTYPE_BF16) return HAN_E_UNSUPPORTED;
    if (in_drop < 0.f || in_drop >= 1.f) return HAN_E_BADARG;
    if (workspace_bytes < han_project_bwd_workspace(N, F, K, FP)) return HAN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int ftiles; int64_t rpc, nch;
    bwd_geometry(N, F, &ftiles, &rpc, &nch);
    ProjBwdArgs a;
    a.X = X; a.x_bf16 = x_dtype == HAN_DTYPE_BF16; a.ldx = ldx; a.dH = dH; a.slab = (float *)workspace; a.N = N; a.F = F;
    a.rows_per_chunk = rpc;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_dev = seed_dev;
    a.thr_in = in_drop > 0.f ? han_keep_threshold(1.f - in_drop) : HAN_KEEP_ALL;
    a.inv_keep_in = 1.f / (1.f - in_drop);
    a.row_offset = row_offset;
    dim3 grid(ftiles, (unsigned)nch);
    const bool vec = !a.x_bf16 && (F % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)X & 15) == 0);
    HAN_DISPATCH_FP(FP, {
        if (in_drop > 0.f) {
            if (vec) project_bwd_kernel<FPC, true, kBwdMT, true><<<grid, 256, 0, st>>>(a);
            else project_bwd_kernel<FPC, true, kBwdMT, false><<<grid, 256, 0, st>>>(a);
        } else {
            if (vec) project_bwd_kernel<FPC, false, kBwdMT, true><<<grid, 256, 0, st>>>(a);
            else project_bwd_kernel<FPC, false, kBwdMT, false><<<grid, 256, 0, st>>>(a);
        }
    })
    HAN_CHECK_LAUNCH();
    const int width = F * HAN_D;
    hipError_t e = han_reduce_slabs((const float *)workspace, (int)nch, width, width, han_reduce_to(dW, width), st);
    if (e != hipSuccess) return (int)e;
    return 0;
}

extern "C" int han_project_bwd_input(const float *dH, const float *W, float *dX, int64_t ldo, int64_t N,
                                     int F, int K, int FP, float in_drop, uint64_t seed,
                                     const uint64_t *seed_dev, int64_t row_offset, void *stream) {
    if (!dH || !W || !dX || N < 0 || F <= 0 || ldo < F) return HAN_E_BADARG;
    if (!fp_ok(K, FP)) return HAN_E_UNSUPPORTED;
    if (in_drop < 0.f || in_drop >= 1.f) return HAN_E_BADARG;
    if (N == 0) return 0;
    ProjBwdInArgs a;
    a.dH = dH; a.W = W; a.dX = dX; a.ldo = ldo; a.N = N; a.F = F;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_dev = seed_dev;
    a.thr_in = in_drop > 0.f ? han_keep_threshold(1.f - in_drop) : HAN_KEEP_ALL;
    a.inv_keep_in = 1.f / (1.f - in_drop);
    a.row_offset = row_offset;
    hipStream_t st = (hipStream_t)stream;
    const int grid = han_grid_for((N + 15) / 16, 4, 256 * 8);
    HAN_DISPATCH_FP(FP, {
        if (in_drop > 0.f) project_bwd_input_kernel<FPC, true><<<grid, 256, 0, st>>>(a);
        else project_bwd_input_kernel<FPC, false><<<grid, 256, 0, st>>>(a);
    })
    HAN_CHECK_LAUNCH();
    return 0;
}
