// Classifier + masked loss, optimiser, and the bias-matrix -> CSR input step (gfx950).
//
// Reference arithmetic:
//   models/gat.py:65-72        logits = (1/HC) sum_h (Z Wc[h] + bc[h])
//   models/base_gattn.py:41-48 masked softmax cross-entropy
//   models/base_gattn.py:61-69 masked accuracy
//   models/base_gattn.py:12-24 L2 on every trainable + tf.train.AdamOptimizer
//   utils/process.py:14-25     additive bias matrix (edge <=> entry > -1e8)
// All of it is row-local streaming work (HBM-bound, tiny next to K2).
#include "han_common.h"

namespace {

constexpr int MAXC = 16;
constexpr int kClsBlocks = 2048;   // 8 waves per SIMD: the per-row chain (load, dot, reduce, exp) is latency-bound

// slab row: [64*C] dWm | [C] dbm | loss | acc
struct ClsArgs {
    const float *Z, *Wc, *bc;
    const int32_t *labels;
    const uint8_t *mask;
    float row_weight;
    float *logits, *dZ, *slab;
    int64_t N;
    int C, HC;
};

// CM = compile-time bound on the class count (4, 8 or 16): every per-class loop is
// unrolled to CM, so small C does not pay for 16 shuffle reductions per row.
// NV = float4 column groups per lane: the embedding width is D = 64*NV (lane q of a 16-lane
// group owns columns 64v + 4q .. 4q+3, v < NV); NV = 2 serves the 128-wide embeddings.
template <bool BWD, int CM, int NV>
__global__ __launch_bounds__(256) void classifier_kernel(const ClsArgs a) {
    constexpr int D = 64 * NV;
    constexpr int NT = 4 * NV;        // features per lane
    __shared__ float Wm[D * MAXC];
    __shared__ float bm[MAXC];
    const int C = a.C;
    const float invh = 1.f / (float)a.HC;
    for (int i = threadIdx.x; i < D * C; i += 256) {
        float s = 0.f;
        for (int h = 0; h < a.HC; ++h) s += a.Wc[(int64_t)h * D * C + i];
        Wm[i] = s * invh;
    }
    if (threadIdx.x < C) {
        float s = 0.f;
        for (int h = 0; h < a.HC; ++h) s += a.bc[h * C + threadIdx.x];
        bm[threadIdx.x] = s * invh;
    }
    __syncthreads();
    const int q = threadIdx.x & 15;
    const int64_t grp0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t ngrp = (int64_t)gridDim.x * 16;
    float dW[NT][CM];
    float dbacc[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        dbacc[c] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) dW[t][c] = 0.f;
    }
    float loss_acc = 0.f, acc_acc = 0.f;
    // this lane's rows of the averaged classifier matrix and the biases, in registers;
    // feature of slot t: 64*(t/4) + 4*q + t%4
    float wq[NT][CM], bq[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        bq[c] = c < C ? bm[c] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) wq[t][c] = c < C ? Wm[(64 * (t >> 2) + 4 * q + (t & 3)) * C + c] : 0.f;
    }
    for (int64_t row = grp0; row < a.N; row += ngrp) {
        float z[NT];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4_t z4 = *reinterpret_cast<const float4_t *>(a.Z + row * D + 64 * v + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) z[4 * v + t] = z4[t];
        }
        float lg[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) s += z[t] * wq[t][c];
            s = han_row16_sum(s);
            lg[c] = c < C ? s + bq[c] : HAN_NEG_BIG;
        }
        if (q < C) {
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < CM; ++c) v = (q == c) ? lg[c] : v;
            a.logits[row * C + q] = v;
        }
        float mx = lg[0];
        int am = 0;
#pragma unroll
        for (int c = 1; c < CM; ++c)
            if (lg[c] > mx) { mx = lg[c]; am = c; }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c) se += c < C ? __expf(lg[c] - mx) : 0.f;
        const int lab = a.labels[row];
        const float w = a.mask[row] ? a.row_weight : 0.f;
        float llab = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c) llab = (c == lab) ? lg[c] : llab;
        if (q == 0) {
            loss_acc += w * (mx + __logf(se) - llab);
            acc_acc += w * (am == lab ? 1.f : 0.f);
        }
        if (BWD) {
            const float inv = 1.f / se;
            float dl[CM];
            float dz[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) dz[t] = 0.f;
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                dl[c] = c < C ? w * (__expf(lg[c] - mx) * inv - (c == lab ? 1.f : 0.f)) : 0.f;
                if (c < C) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        dz[t] += dl[c] * wq[t][c];
                        dW[t][c] += z[t] * dl[c];
                    }
                    if (q == 0) dbacc[c] += dl[c];
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float4_t o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = dz[4 * v + t];
                *reinterpret_cast<float4_t *>(a.dZ + row * D + 64 * v + 4 * q) = o;
            }
        }
    }
    // block reduction: 16 groups -> one slab row.
    __syncthreads();
    float *scr = Wm;                  // [D*C]
    __shared__ float sc2[MAXC + 2];   // db | loss | acc
    const int grp = threadIdx.x >> 4;
    // Small class counts (the 16 groups' partial rows fit 33 KB of LDS): every group writes its row, ONE barrier,
    // every thread adds the 16 partials of its columns in group order -- instead of 16 barrier rounds through one
    // scratch row (at the size of the reference's data sets those rounds were half of this kernel's 14 us).
    constexpr int PW = D * CM + CM + 2;                 // dW | db | loss | acc of one group
    constexpr bool PAR = PW * 16 * 4 <= 34 * 1024;
    __shared__ float part[PAR ? 16 * PW : 1];
    if (PAR) {
        float *mine = part + grp * PW;
        if (BWD) {
#pragma unroll
            for (int c = 0; c < CM; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t) mine[(64 * (t >> 2) + 4 * q + (t & 3)) * CM + c] = dW[t][c];
            if (q == 0) {
#pragma unroll
                for (int c = 0; c < CM; ++c) mine[D * CM + c] = dbacc[c];
            }
        }
        if (q == 0) {
            mine[D * CM + CM] = loss_acc;
            mine[D * CM + CM + 1] = acc_acc;
        }
        __syncthreads();
        const int width = D * C + C + 2;
        float *out = a.slab + (int64_t)blockIdx.x * width;
        for (int i = threadIdx.x; i < width; i += 256) {
            int src;                                        // column i of the slab row -> its slot in a group's partial row
            if (i < D * C) src = (i / C) * CM + (i % C);
            else if (i < D * C + C) src = D * CM + (i - D * C);
            else src = D * CM + CM + (i - D * C - C);
            float v = 0.f;
            if (BWD || i >= D * C + C) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v += part[r * PW + src];
            }
            out[i] = v;
        }
        return;
    }
    // larger class bounds: serialise the groups through Wm-sized scratch (one group adds at a time; 16 barriers)
    for (int r = 0; r < 16; ++r) {
        if (grp == r) {
            if (BWD) {
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (c < C) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const int idx = (64 * (t >> 2) + 4 * q + (t & 3)) * C + c;
                            scr[idx] = (r == 0 ? 0.f : scr[idx]) + dW[t][c];
                        }
                        if (q == 0) sc2[c] = (r == 0 ? 0.f : sc2[c]) + dbacc[c];
                    }
                }
            }
            if (q == 0) {
                sc2[MAXC] = (r == 0 ? 0.f : sc2[MAXC]) + loss_acc;
                sc2[MAXC + 1] = (r == 0 ? 0.f : sc2[MAXC + 1]) + acc_acc;
            }
        }
        __syncthreads();
    }
    const int width = D * C + C + 2;
    float *out = a.slab + (int64_t)blockIdx.x * width;
    for (int i = threadIdx.x; i < width; i += 256) {
        float v;
        if (i < D * C) v = BWD ? scr[i] : 0.f;
        else if (i < D * C + C) v = BWD ? sc2[i - D * C] : 0.f;
        else v = sc2[MAXC + (i - D * C - C)];
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Many classes (16 < C <= 64): one wave per row, lane c = class c.  The averaged classifier column of
// the lane's class and its gradient accumulator live in registers (2*D VGPRs); the row z is broadcast
// element by element with v_readlane (lane d holds z[d], z[d + 64]); softmax statistics are wave
// reductions; dZ is formed by lane d from the LDS copy of the matrix (row stride C_LD = 65: lanes that
// step d hit different banks).  ~ (2D + C) * 3 vector instructions per row -- several times the cost
// per row of the small-C kernel, which keeps 16 rows per block iteration; only used when C > 16.
// ---------------------------------------------------------------------------------------------
constexpr int WC_LD = 65;

template <bool BWD, int NV>
__global__ __launch_bounds__(256) void classifier_wide_kernel(const ClsArgs a) {
    constexpr int D = 64 * NV;
    __shared__ float Wl[D * WC_LD];       // averaged matrix [d][c]; reused as the reduction scratch at the end
    __shared__ float sc2[64 + 2];         // db | loss | acc
    const int C = a.C;
    const float invh = 1.f / (float)a.HC;
    for (int i = threadIdx.x; i < D * C; i += 256) {
        float s = 0.f;
        for (int h = 0; h < a.HC; ++h) s += a.Wc[(int64_t)h * D * C + i];
        Wl[(i / C) * WC_LD + (i % C)] = s * invh;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool cls = lane < C;
    float bq = 0.f;
    if (cls)
        for (int h = 0; h < a.HC; ++h) bq += a.bc[h * C + lane];
    bq *= invh;
    float wcol[D], dwcol[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        wcol[d] = cls ? Wl[d * WC_LD + lane] : 0.f;
        dwcol[d] = 0.f;
    }
    float dbacc = 0.f, loss_acc = 0.f, acc_acc = 0.f;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < a.N; row += nwaves) {
        float z[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) z[v] = a.Z[row * D + 64 * v + lane];
        float lg = bq;
#pragma unroll
        for (int d = 0; d < D; ++d)
            lg += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[d >> 6]), d & 63)) * wcol[d];
        if (cls) a.logits[row * C + lane] = lg;
        const float lgm = cls ? lg : HAN_NEG_BIG;
        float mx = lgm;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const unsigned long long hit = __ballot(cls && lgm == mx);
        const int am = __ffsll((long long)hit) - 1;              // first maximum, as argmax
        const float ex = cls ? __expf(lg - mx) : 0.f;
        const float se = han_wave_sum(ex);
        const int lab = a.labels[row];
        const float wgt = a.mask[row] ? a.row_weight : 0.f;
        const float llab = __shfl(lg, lab, 64);
        if (lane == 0) {
            loss_acc += wgt * (mx + __logf(se) - llab);
            acc_acc += wgt * (am == lab ? 1.f : 0.f);
        }
        if (BWD) {
            const float dl = cls ? wgt * (ex / se - (lane == lab ? 1.f : 0.f)) : 0.f;
            dbacc += dl;
#pragma unroll
            for (int d = 0; d < D; ++d)
                dwcol[d] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[d >> 6]), d & 63)) * dl;
            float dz[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) dz[v] = 0.f;
            for (int c = 0; c < C; ++c) {
                const float dlc = __shfl(dl, c, 64);
#pragma unroll
                for (int v = 0; v < NV; ++v) dz[v] += dlc * Wl[(64 * v + lane) * WC_LD + c];
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) a.dZ[row * D + 64 * v + lane] = dz[v];
        }
    }
    // block reduction: the 4 waves add their dW columns one after the other into the LDS matrix
    __syncthreads();
    for (int ww = 0; ww < 4; ++ww) {
        if (w == ww) {
            if (BWD && cls) {
#pragma unroll
                for (int d = 0; d < D; ++d) Wl[d * WC_LD + lane] = (ww == 0 ? 0.f : Wl[d * WC_LD + lane]) + dwcol[d];
                sc2[lane] = (ww == 0 ? 0.f : sc2[lane]) + dbacc;
            }
            if (lane == 0) {
                sc2[64] = (ww == 0 ? 0.f : sc2[64]) + loss_acc;
                sc2[65] = (ww == 0 ? 0.f : sc2[65]) + acc_acc;
            }
        }
        __syncthreads();
    }
    const int width = D * C + C + 2;
    float *out = a.slab + (int64_t)blockIdx.x * width;
    for (int i = threadIdx.x; i < width; i += 256) {
        float v;
        if (i < D * C) v = BWD ? Wl[(i / C) * WC_LD + (i % C)] : 0.f;
        else if (i < D * C + C) v = BWD ? sc2[i - D * C] : 0.f;
        else v = sc2[64 + (i - D * C - C)];
        out[i] = v;
    }
}

template <int NV>
static void launch_classifier(const ClsArgs &a, bool bwd, int grid, hipStream_t st) {
    if (a.C <= 4) {
        if (bwd) classifier_kernel<true, 4, NV><<<grid, 256, 0, st>>>(a);
        else classifier_kernel<false, 4, NV><<<grid, 256, 0, st>>>(a);
    } else if (a.C <= 8) {
        if (bwd) classifier_kernel<true, 8, NV><<<grid, 256, 0, st>>>(a);
        else classifier_kernel<false, 8, NV><<<grid, 256, 0, st>>>(a);
    } else {
        if (bwd) classifier_kernel<true, 16, NV><<<grid, 256, 0, st>>>(a);
        else classifier_kernel<false, 16, NV><<<grid, 256, 0, st>>>(a);
    }
}

__global__ void adam_kernel(float *p, const float *g, float *m, float *v, int64_t n, float lr_t, float b1,
                            float b2, float eps, float l2, const int64_t *step_dev) {
    if (step_dev) {   // lr_t holds the base rate; the bias correction comes from the device step count
        const double t = (double)*step_dev;
        lr_t = (float)((double)lr_t * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float pi = p[i];
        const float gi = g[i] + l2 * pi;             // d/dp of l2_coef * p^2/2
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - lr_t * mi / (sqrtf(vi) + eps);
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float *p, int64_t n, float *partial) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += p[i] * p[i];
    s = han_wave_sum(s);
    __shared__ float w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ void sumsq_finish_kernel(const float *partial, int nblocks, float *out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) s += partial[i];
    s = han_wave_sum(s);
    if (threadIdx.x == 0) out[0] = 0.5f * s;
}

// one wave per row of the dense bias matrix
__global__ __launch_bounds__(256) void bias_count_kernel(const float *bias, int64_t N, int64_t ld,
                                                         int64_t *counts) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    int cnt = 0;
    for (int64_t c0 = 0; c0 < N; c0 += 64) {
        const int64_t c = c0 + lane;
        const bool edge = c < N && bias[row * ld + c] > -1e8f;
        cnt += __popcll(__ballot(edge));
    }
    if (lane == 0) counts[row] = cnt;
}

__global__ __launch_bounds__(256) void bias_fill_kernel(const float *bias, int64_t N, int64_t ld,
                                                        const int64_t *rowptr, int32_t *colidx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    int64_t pos = rowptr[row];
    for (int64_t c0 = 0; c0 < N; c0 += 64) {
        const int64_t c = c0 + lane;
        const bool edge = c < N && bias[row * ld + c] > -1e8f;
        const unsigned long long b = __ballot(edge);
        if (edge) colidx[pos + __popcll(b & ((1ull << lane) - 1ull))] = (int32_t)c;
        pos += __popcll(b);
    }
}


// ---------------------------------------------------------------------------------------------
// Any embedding width (D a multiple of 64: a last layer of 8 heads x 32, hid_units = [128] ...) and any
// class count (round 3; models/gat.py:61-72 leaves both free).  Three small kernels instead of one:
//   cls_mean_kernel   WmT[c][d] = mean_h Wc[h][d][c], bm[c] = mean_h bc[h][c]   (the head average, once)
//   cls_gen_kernel    one wave per row, lane = feature: logit_c = wave_sum(z . WmT[c]) + bm[c] class by class
//                     (online max / sum / first argmax), loss + accuracy; backward: dl_c, dZ = sum_c dl_c WmT[c]
//                     in registers, dl rows to a scratch table.  Rows outside the loss mask have dl == 0
//                     exactly: their dZ row is written as zeros and the class loop is skipped.
//   cls_dw_kernel     dWm = Z^T dL over the MASKED rows only (the others add exactly 0), one block per
//                     (row split, 64-feature tile, 16-class tile), lane = feature; db from the feature-tile-0 blocks
// NV = registers holding the row (64*NV >= D), or 0: any width, the row is re-read from L1 per class.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cls_mean_kernel(const float *Wc, const float *bc, float *WmT, float *bm,
                                                       int D, int C, int HC) {
    const float invh = 1.f / (float)HC;
    const int64_t n = (int64_t)D * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i / D), d = (int)(i % D);
        float s = 0.f;
        for (int h = 0; h < HC; ++h) s += Wc[(int64_t)h * n + (int64_t)d * C + c];
        WmT[i] = s * invh;
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.f;
            for (int h = 0; h < HC; ++h) s += bc[h * C + c];
            bm[c] = s * invh;
        }
}

struct ClsGenArgs {
    const float *Z, *WmT, *bm;
    const int32_t *labels;
    const uint8_t *mask;
    float row_weight;
    float *logits, *dZ, *dL, *slab;    // slab: [gridDim.x][2] loss | acc
    int64_t N;
    int D, C;
};

template <bool BWD, int NV>
__global__ __launch_bounds__(256) void cls_gen_kernel(const ClsGenArgs a) {
    constexpr int NR = NV > 0 ? NV : 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int D = a.D, C = a.C, nv = D / 64;
    float loss_acc = 0.f, acc_acc = 0.f;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < a.N; row += nwaves) {
        const float *zrow = a.Z + row * D + lane;
        float z[NR];
        if (NV > 0) {
#pragma unroll
            for (int v = 0; v < NR; ++v) z[v] = v < nv ? zrow[64 * v] : 0.f;
        }
        float mx = HAN_NEG_BIG, se = 0.f, llab = 0.f;
        int am = 0;
        const int lab = a.labels[row];
        const bool live = a.mask[row] != 0;      // the predicate cls_dw_kernel skips rows by: a live row ALWAYS gets its dL row
        const float wgt = live ? a.row_weight : 0.f;
        for (int c = 0; c < C; ++c) {
            const float *wr = a.WmT + (int64_t)c * D + lane;
            float part = 0.f;
            if (NV > 0) {
#pragma unroll
                for (int v = 0; v < NR; ++v) part += v < nv ? z[v] * wr[64 * v] : 0.f;
            } else {
                for (int v = 0; v < nv; ++v) part += zrow[64 * v] * wr[64 * v];
            }
            const float lg = han_wave_sum(part) + a.bm[c];
            if (lane == 0) a.logits[row * C + c] = lg;
            if (lg > mx) {       // first maximum, as argmax
                se = se * __expf(mx - lg) + 1.f;
                mx = lg;
                am = c;
            } else {
                se += __expf(lg - mx);
            }
            llab = (c == lab) ? lg : llab;
        }
        if (lane == 0) {
            loss_acc += wgt * (mx + __logf(se) - llab);
            acc_acc += wgt * (am == lab ? 1.f : 0.f);
        }
        if (BWD) {
            float dz[NR];
#pragma unroll
            for (int v = 0; v < NR; ++v) dz[v] = 0.f;
            if (live) {              // wave-uniform: rows outside the mask have dl == 0 exactly (a live row with
                                     // row_weight == 0 writes zeros: cls_dw_kernel reads the dL row of every live row)
                const float inv = 1.f / se;
                if (NV == 0)
                    for (int v = 0; v < nv; ++v) a.dZ[row * D + 64 * v + lane] = 0.f;
                for (int c = 0; c < C; ++c) {
                    // lane 0 wrote the logit; it reads it back (same thread) and hands it to the wave
                    const float lg = __int_as_float(__builtin_amdgcn_readfirstlane(
                        __float_as_int(lane == 0 ? a.logits[row * C + c] : 0.f)));
                    const float dl = wgt * (__expf(lg - mx) * inv - (c == lab ? 1.f : 0.f));
                    if (lane == 0) a.dL[row * C + c] = dl;
                    const float *wr = a.WmT + (int64_t)c * D + lane;
                    if (NV > 0) {
#pragma unroll
                        for (int v = 0; v < NR; ++v) dz[v] += v < nv ? dl * wr[64 * v] : 0.f;
                    } else {
                        for (int v = 0; v < nv; ++v) a.dZ[row * D + 64 * v + lane] += dl * wr[64 * v];
                    }
                }
            }
            if (NV > 0 || !live) {
                if (NV > 0) {
#pragma unroll
                    for (int v = 0; v < NR; ++v)
                        if (v < nv) a.dZ[row * D + 64 * v + lane] = dz[v];
                } else {
                    for (int v = 0; v < nv; ++v) a.dZ[row * D + 64 * v + lane] = 0.f;
                }
            }
        }
    }
    __shared__ float red[4][2];
    if (lane == 0) { red[w][0] = loss_acc; red[w][1] = acc_acc; }
    __syncthreads();
    if (threadIdx.x < 2)
        a.slab[(int64_t)blockIdx.x * 2 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// slab row per row split: [D*C] dWm | [C] dbm.  grid = (splits, D/64 feature tiles, ceil(C/16) class tiles)
__global__ __launch_bounds__(256) void cls_dw_kernel(const float *Z, const float *dL, const uint8_t *mask, float *slab,
                                                     int64_t N, int D, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int d0 = 64 * blockIdx.y, c0 = 16 * blockIdx.z;
    const int nc = C - c0 < 16 ? C - c0 : 16;
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = per * blockIdx.x, r1 = (r0 + per < N) ? r0 + per : N;
    float acc[16], dbacc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[j] = 0.f; dbacc[j] = 0.f; }
    for (int64_t row = r0 + w; row < r1; row += 4) {
        if (mask && !mask[row]) continue;               // wave-uniform; dl of such a row is exactly 0
        const float z = Z[row * D + d0 + lane];
        const float *dl = dL + row * C + c0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float v = j < nc ? dl[j] : 0.f;
            acc[j] += z * v;
            dbacc[j] += v;
        }
    }
    __shared__ float red[4][16][65];
#pragma unroll
    for (int j = 0; j < 16; ++j) red[w][j][lane] = acc[j];
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) red[w][j][64] = dbacc[j];
    }
    __syncthreads();
    float *out = slab + (int64_t)blockIdx.x * ((int64_t)D * C + C);
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int j = i & 15, d = i >> 4;
        if (j < nc) out[(int64_t)(d0 + d) * C + c0 + j] = red[0][j][d] + red[1][j][d] + red[2][j][d] + red[3][j][d];
    }
    if (blockIdx.y == 0 && threadIdx.x < nc)
        out[(int64_t)D * C + c0 + threadIdx.x] = red[0][threadIdx.x][64] + red[1][threadIdx.x][64] + red[2][threadIdx.x][64] + red[3][threadIdx.x][64];
}

// dZ = dL . Wm^T for a given dL (the backward of the logits alone, HeteGAT_multi.inference without the fused loss)
__global__ __launch_bounds__(256) void cls_dz_kernel(const float *dL, const float *WmT, float *dZ, int64_t N, int D, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < N; row += nwaves) {
        const float *dl = dL + row * C;
        for (int f = lane; f < D; f += 64) {
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += dl[c] * WmT[(int64_t)c * D + f];
            dZ[row * D + f] = s;
        }
    }
}

constexpr int kClsGenBlocks = 2048;

static int cls_dw_splits(int D, int C) {
    const int tiles = (D / 64) * ((C + 15) / 16);
    int s = 1024 / tiles;
    return s < 1 ? 1 : (s > 256 ? 256 : s);
}

// workspace of the general path, in floats: WmT | bm | loss/acc slabs | dL | dW slabs
static size_t cls_gen_workspace_floats(int64_t N, int D, int C) {
    return (size_t)D * C + (size_t)C + (size_t)kClsGenBlocks * 2 + (size_t)N * C +
           (size_t)cls_dw_splits(D, C) * ((size_t)D * C + C);
}

static bool cls_tuned(int D, int C) { return (D == 64 || D == 128) && C >= 1 && C <= 64; }

}  // namespace

extern "C" size_t han_classifier_workspace(int64_t N, int D, int C, int HC) {
    (void)HC;
    if (D >= 64 && D % 64 == 0 && C >= 1 && !cls_tuned(D, C)) return cls_gen_workspace_floats(N, D, C) * sizeof(float);
    return (size_t)kClsBlocks * (size_t)(D * C + C + 2) * sizeof(float);
}

extern "C" int han_classifier_loss(const float *Z, const float *Wc, const float *bc, const int32_t *labels,
                                   const uint8_t *mask, float row_weight, float *logits, float *loss_acc,
                                   float *dZ, float *dWc, float *dbc, void *workspace, size_t workspace_bytes,
                                   int64_t N, int D, int C, int HC, void *stream) {
    if (!Z || !Wc || !bc || !labels || !mask || !logits || !loss_acc || !workspace || N < 0 || HC <= 0)
        return HAN_E_BADARG;
    if (D < 64 || D % 64 != 0 || C < 1) return HAN_E_UNSUPPORTED;
    const bool bwd = dZ != nullptr;
    if (bwd && (!dWc || !dbc)) return HAN_E_BADARG;
    if (workspace_bytes < han_classifier_workspace(N, D, C, HC)) return HAN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!cls_tuned(D, C)) {       // any embedding width / class count: the three-kernel path
        float *ws = (float *)workspace;
        ClsGenArgs g;
        float *WmT = ws, *bm = WmT + (size_t)D * C, *la = bm + C, *dL = la + (size_t)kClsGenBlocks * 2,
              *dws = dL + (size_t)N * C;
        g.Z = Z; g.WmT = WmT; g.bm = bm; g.labels = labels; g.mask = mask; g.row_weight = row_weight;
        g.logits = logits; g.dZ = dZ; g.dL = dL; g.slab = la; g.N = N; g.D = D; g.C = C;
        cls_mean_kernel<<<han_grid_for((int64_t)D * C, 256, 256), 256, 0, st>>>(Wc, bc, WmT, bm, D, C, HC);
        HAN_CHECK_LAUNCH();
        const int grid = han_grid_for(N > 0 ? N : 1, 4, kClsGenBlocks);
#define HAN_CLS_GEN(NVV)                                                   \
    if (bwd) cls_gen_kernel<true, NVV><<<grid, 256, 0, st>>>(g);           \
    else cls_gen_kernel<false, NVV><<<grid, 256, 0, st>>>(g);
        if (D <= 256) { HAN_CLS_GEN(4) } else if (D <= 512) { HAN_CLS_GEN(8) } else if (D <= 1024) { HAN_CLS_GEN(16) } else { HAN_CLS_GEN(0) }
#undef HAN_CLS_GEN
        HAN_CHECK_LAUNCH();
        hipError_t e = han_reduce_slabs(la, grid, 2, 2, han_reduce_to(loss_acc, 2), st);
        if (e != hipSuccess) return (int)e;
        if (bwd) {
            const int S = cls_dw_splits(D, C);
            cls_dw_kernel<<<dim3(S, D / 64, (C + 15) / 16), 256, 0, st>>>(Z, dL, mask, dws, N, D, C);
            HAN_CHECK_LAUNCH();
            const int width = D * C + C;
            HanReduceOut o = han_reduce_to(dWc, width);
            o.nseg = 2;
            o.ptr[1] = dbc;
            o.seg_end[0] = D * C; o.seg_end[1] = width;
            o.scale[0] = o.scale[1] = 1.f / (float)HC;
            o.rep[0] = o.rep[1] = HC;
            o.rep_stride[0] = (int64_t)D * C; o.rep_stride[1] = C;
            e = han_reduce_slabs(dws, S, width, width, o, st);
            if (e != hipSuccess) return (int)e;
        }
        return 0;
    }
    ClsArgs a;
    a.Z = Z; a.Wc = Wc; a.bc = bc; a.labels = labels; a.mask = mask; a.row_weight = row_weight;
    a.logits = logits; a.dZ = dZ; a.slab = (float *)workspace; a.N = N; a.C = C; a.HC = HC;
    int grid = han_grid_for(N > 0 ? N : 1, 16, kClsBlocks);
    if (C > MAXC) {        // class-per-lane kernel, one wave per row
        grid = han_grid_for(N > 0 ? N : 1, 4, kClsBlocks);
        if (D == 128) {
            if (bwd) classifier_wide_kernel<true, 2><<<grid, 256, 0, st>>>(a);
            else classifier_wide_kernel<false, 2><<<grid, 256, 0, st>>>(a);
        } else {
            if (bwd) classifier_wide_kernel<true, 1><<<grid, 256, 0, st>>>(a);
            else classifier_wide_kernel<false, 1><<<grid, 256, 0, st>>>(a);
        }
    } else if (D == 128) launch_classifier<2>(a, bwd, grid, st);
    else launch_classifier<1>(a, bwd, grid, st);
    HAN_CHECK_LAUNCH();
    const int width = D * C + C + 2;
    // one second-stage launch: the head gradients (every head receives the same (1/HC)-scaled gradient,
    // models/gat.py:72 averages them) and loss / accuracy (the last two slab columns)
    const float *slab = (const float *)workspace;
    if (bwd) {
        HanReduceOut o = han_reduce_to(dWc, width);
        o.nseg = 3;
        o.ptr[1] = dbc; o.ptr[2] = loss_acc;
        o.seg_end[0] = D * C; o.seg_end[1] = D * C + C; o.seg_end[2] = width;
        o.scale[0] = o.scale[1] = 1.f / (float)HC;
        o.rep[0] = o.rep[1] = HC;
        o.rep_stride[0] = (int64_t)D * C; o.rep_stride[1] = C;
        hipError_t e = han_reduce_slabs(slab, grid, width, width, o, st);
        if (e != hipSuccess) return (int)e;
    } else {
        hipError_t e = han_reduce_slabs(slab + D * C + C, grid, width, 2, han_reduce_to(loss_acc, 2), st);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

extern "C" size_t han_classifier_bwd_workspace(int64_t N, int D, int C, int HC) {
    (void)N; (void)HC;
    if (D < 64 || D % 64 != 0 || C < 1) return 0;
    return ((size_t)D * C + (size_t)C + (size_t)cls_dw_splits(D, C) * ((size_t)D * C + C)) * sizeof(float);
}

extern "C" int han_classifier_bwd(const float *Z, const float *Wc, const float *bc, const float *dlogits, float *dZ,
                                  float *dWc, float *dbc, void *workspace, size_t workspace_bytes, int64_t N, int D,
                                  int C, int HC, void *stream) {
    if (!Z || !Wc || !bc || !dlogits || !dZ || !dWc || !dbc || !workspace || N < 0 || HC <= 0) return HAN_E_BADARG;
    if (D < 64 || D % 64 != 0 || C < 1) return HAN_E_UNSUPPORTED;
    if (workspace_bytes < han_classifier_bwd_workspace(N, D, C, HC)) return HAN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *WmT = (float *)workspace, *bm = WmT + (size_t)D * C, *dws = bm + C;
    cls_mean_kernel<<<han_grid_for((int64_t)D * C, 256, 256), 256, 0, st>>>(Wc, bc, WmT, bm, D, C, HC);
    HAN_CHECK_LAUNCH();
    cls_dz_kernel<<<han_grid_for(N > 0 ? N : 1, 4, kClsGenBlocks), 256, 0, st>>>(dlogits, WmT, dZ, N, D, C);
    HAN_CHECK_LAUNCH();
    const int S = cls_dw_splits(D, C);
    cls_dw_kernel<<<dim3(S, D / 64, (C + 15) / 16), 256, 0, st>>>(Z, dlogits, nullptr, dws, N, D, C);
    HAN_CHECK_LAUNCH();
    const int width = D * C + C;
    HanReduceOut o = han_reduce_to(dWc, width);
    o.nseg = 2;
    o.ptr[1] = dbc;
    o.seg_end[0] = D * C; o.seg_end[1] = width;
    o.scale[0] = o.scale[1] = 1.f / (float)HC;
    o.rep[0] = o.rep[1] = HC;
    o.rep_stride[0] = (int64_t)D * C; o.rep_stride[1] = C;
    hipError_t e = han_reduce_slabs(dws, S, width, width, o, st);
    if (e != hipSuccess) return (int)e;
    return 0;
}

extern "C" int han_adam_step(float *param, const float *grad, float *m, float *v, int64_t n, float lr_t,
                             float beta1, float beta2, float eps, float l2_coef, const int64_t *step_dev,
                             void *stream) {
    if (!param || !grad || !m || !v || n < 0) return HAN_E_BADARG;
    if (n == 0) return 0;
    adam_kernel<<<han_grid_for(n, 256, 2048), 256, 0, (hipStream_t)stream>>>(param, grad, m, v, n, lr_t, beta1,
                                                                            beta2, eps, l2_coef, step_dev);
    HAN_CHECK_LAUNCH();
    return 0;
}

extern "C" int han_l2_half_sumsq(const float *param, int64_t n, float *out, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    if (!param || !out || !workspace || n < 0) return HAN_E_BADARG;
    if (workspace_bytes < 4096 * sizeof(float)) return HAN_E_WORKSPACE;
    const int grid = han_grid_for(n > 0 ? n : 1, 256, 1024);
    sumsq_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(param, n, (float *)workspace);
    HAN_CHECK_LAUNCH();
    sumsq_finish_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const float *)workspace, grid, out);
    HAN_CHECK_LAUNCH();
    return 0;
}

extern "C" int han_bias_row_counts(const float *bias, int64_t N, int64_t ld, int64_t *counts, void *stream) {
    if (!bias || !counts || N < 0 || ld < N) return HAN_E_BADARG;
    if (N == 0) return 0;
    bias_count_kernel<<<(unsigned)((N + 3) / 4), 256, 0, (hipStream_t)stream>>>(bias, N, ld, counts);
    HAN_CHECK_LAUNCH();
    return 0;
}

extern "C" int han_bias_fill_csr(const float *bias, int64_t N, int64_t ld, const int64_t *rowptr,
                                 int32_t *colidx, void *stream) {
    if (!bias || !rowptr || !colidx || N < 0 || ld < N) return HAN_E_BADARG;
    if (N == 0) return 0;
    bias_fill_kernel<<<(unsigned)((N + 3) / 4), 256, 0, (hipStream_t)stream>>>(bias, N, ld, rowptr, colidx);
    HAN_CHECK_LAUNCH();
    return 0;
}

extern "C" int han_abi_version(void) { return HAN_ABI_VERSION; }

extern "C" const char *han_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case HAN_E_BADARG: return "han: bad argument (null pointer, negative size or inconsistent shape)";
        case HAN_E_UNSUPPORTED: return "han: shape not supported by this build (K*FP == 64 with FP in {4,8,16,32,64}; D and A multiples of 64)";
        case HAN_E_WORKSPACE: return "han: workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "han: unknown error";
    }
}
