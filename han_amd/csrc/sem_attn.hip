// K3 -- semantic-level (meta-path) attention, forward and backward (gfx950).
//
// Reference arithmetic: utils/layers.py:152-159 (SimpleAttLayer):
//   v = tanh(M @ Womega + bomega)   (N,P,A)      s = v . uomega   (N,P)
//   beta = softmax over P, PER NODE (the code, not the paper's node average)
//   Z = sum_p beta_p M_p            (N,D)
// M is (N,P,64) contiguous = R = N*P rows of 256 B.  v (2 GB at N = 1M, P = 4)
// is never written.  The D x A contraction runs on exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32, no xf32 on gfx950); everything around it (tanh, the
// score dot, the per-node softmax, the weighted sum) is fused in the same kernel.
//
// Forward, per block iteration: a chunk of NB = 64/P whole nodes (<= 64 rows);
// wave w owns rows 16w..16w+15: pre = M_tile(16x64) . Womega(64xA) with the M
// fragments loaded straight from global/L1 and Womega fragments from LDS; the
// scores go to LDS; then each wave finishes whole nodes (lane = feature: softmax
// over P scores, Z = sum beta_p M_p with 256-B coalesced row loads).
//
// Backward, per 16-row tile and wave, three MFMA products of 128 MFMAs each:
//   G1  pre   = M_tile . Womega                       (recompute; rows r x cols a)
//   G3  dW   += M_tile^T . dpre      reduction over r = the ROW index of G1's
//                                    accumulator, so dpre is fed as the B operand
//                                    straight from the accumulator registers
//   G2  dMx   = dpre . Womega^T      reduction over a -> dpre goes through a
//                                    wave-private LDS tile to become the A operand
// with dpre = ds_r * u * (1 - v^2), ds_p = beta_p (dbeta_p - sum_q beta_q dbeta_q),
// dbeta_p = dZ . M_p.  dW/db/du are summed in registers over the wave's tiles,
// across waves through LDS, across blocks through a slab + reduce kernel.
//
// Roofline: MFMA fp32 (155 TF): forward 2*R*64*A flop = 65.5 GF at N = 1M, P = 4
// (0.42 ms at peak), backward 3x that; the M / Z / dM streams are 1-2 GB (HBM, ~0.3 ms).
#include "han_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2/(1+e^{2x}) with v_exp_f32 + v_rcp_f32 (a plain or "fast" division expands to
    // a ~10-instruction div_scale/div_fmas sequence); absolute error ~2e-7, saturates at +-1
    const float e = __expf(2.f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
}

constexpr int ROWS = 64;   // rows per block iteration (4 waves x 16)

// A node with P = 8 / 16 meta-paths spans 2 / 4 of the 16-lane groups of a wave tile (4 rows
// each): reductions over its rows finish with one / two cross-group exchanges.
template <int P>
__device__ __forceinline__ float node_xsum(float v) {
    if (P >= 8) v += __shfl_xor(v, 16, 64);
    if (P >= 16) v += __shfl_xor(v, 32, 64);
    return v;
}
template <int P>
__device__ __forceinline__ float node_xmax(float v) {
    if (P >= 8) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (P >= 16) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

template <int CA>
__global__ __launch_bounds__(256) void sem_attn_fwd_kernel(const float *__restrict__ M, const float *Wg,
                                                           const float *bw, const float *uw, float *Z,
                                                           float *beta, int64_t N, int P) {
    constexpr int A = 64 * CA;
    constexpr int TA = A / 16;
    constexpr int WLD = A + 16;   // (WLD mod 32) == 16: rows k and k+1 hit disjoint bank halves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;                 // [64][WLD]
    float *sc = smem + 64 * WLD;      // [2][ROWS]
    for (int i = threadIdx.x; i < 64 * A; i += 256) Wl[(i / A) * WLD + (i % A)] = Wg[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    float bcol[TA], ucol[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        bcol[t] = bw[16 * t + l15];
        ucol[t] = uw[16 * t + l15];
    }
    __syncthreads();
    const int NB = ROWS / P;
    const int64_t nchunks = (N + NB - 1) / NB;
    int buf = 0;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x, buf ^= 1) {
        const int64_t node0 = ch * NB;
        const int nodes = (int)((N - node0) < NB ? (N - node0) : NB);
        const int rows = nodes * P;
        const int64_t row0 = node0 * P;
        // ---- scores of this wave's 16 rows
        {
            const int lr = 16 * w + l15;
            const float *mrow = M + (row0 + (lr < rows ? lr : rows - 1)) * 64 + l4;
            f32x4 acc[TA];
#pragma unroll
            for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const float a = mrow[4 * ks];
                const float *wr = Wl + (4 * ks + l4) * WLD + l15;
#pragma unroll
                for (int t = 0; t < TA; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wr[16 * t], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < TA; ++t) s += fast_tanh(acc[t][reg] + bcol[t]) * ucol[t];
                s = han_row16_sum(s);
                if (l15 == 0) sc[buf * ROWS + 16 * w + 4 * l4 + reg] = s;
            }
        }
        __syncthreads();
        // ---- whole nodes: softmax over P, weighted sum (lane = feature)
        for (int nd = w; nd < nodes; nd += 4) {
            const int64_t n = node0 + nd;
            float zacc = 0.f, mrun = HAN_NEG_BIG, lrun = 0.f, sreg = 0.f;
            for (int p = 0; p < P; ++p) {
                const float s = sc[buf * ROWS + nd * P + p];
                const float mval = M[(n * P + p) * 64 + lane];
                if (lane == p) sreg = s;
                const float mn = fmaxf(mrun, s);
                const float scl = __expf(mrun - mn), pe = __expf(s - mn);
                lrun = lrun * scl + pe;
                zacc = zacc * scl + pe * mval;
                mrun = mn;
            }
            const float inv = 1.f / lrun;
            Z[n * 64 + lane] = zacc * inv;
            if (lane < P) beta[n * P + lane] = __expf(sreg - mrun) * inv;
        }
        // no second barrier: the next iteration writes the other score buffer, and
        // the barrier inside it orders this iteration's reads before the reuse after.
    }
}

// Forward for P in {1,2,4} (P divides 4; P = 8, 16 span 2 / 4 groups, see node_xsum): the P rows of a node sit in ONE 16-lane
// group of the accumulator layout (rows 4*l4 .. 4*l4+3), so the per-node softmax
// and the weighted sum finish inside the group -- no LDS exchange, no barrier; each
// wave streams its own 16-row tiles.
template <int CA, int P, bool WREG>
__global__ __launch_bounds__(256) void sem_attn_fwd_wave_kernel(const float *__restrict__ M, const float *Wg,
                                                                const float *bw, const float *uw, float *Z,
                                                                float *beta, int64_t N) {
    constexpr int A = 64 * CA;
    constexpr int TA = A / 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // Womega lives in registers for the whole kernel: B fragment of k-step ks, column
    // tile t is W[4ks + l4][16t + l15]  (16*TA VGPRs; no LDS traffic in the MFMA loop)
    constexpr int WLD = A + 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float wf[WREG ? 16 : 1][TA];
    if (WREG) {
#pragma unroll
        for (int ks = 0; ks < (WREG ? 16 : 1); ++ks)
#pragma unroll
            for (int t = 0; t < TA; ++t) wf[ks][t] = Wg[(4 * ks + l4) * A + 16 * t + l15];
    } else {
        for (int i = threadIdx.x; i < 64 * A; i += 256) smem[(i / A) * WLD + (i % A)] = Wg[i];
        __syncthreads();
    }
    float bcol[TA], ucol[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        bcol[t] = bw[16 * t + l15];
        ucol[t] = uw[16 * t + l15];
    }
    const int64_t R = N * P;
    const int64_t ntiles = (R + 15) / 16;
    const int64_t tstride = (int64_t)gridDim.x * 4;
    float afrag[16];       // A fragments of the current tile, loaded one tile ahead
    {
        const int64_t t0 = (int64_t)blockIdx.x * 4 + w;
        const int64_t ra = t0 * 16 + l15 < R ? t0 * 16 + l15 : R - 1;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) afrag[ks] = M[ra * 64 + l4 + 4 * ks];
    }
    for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += tstride) {
        const int64_t r0 = tile * 16;
        f32x4 acc[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float anext[16];
        {
            const int64_t rn = (tile + tstride) * 16 + l15;
            const float *mnext = M + (rn < R ? rn : R - 1) * 64 + l4;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) anext[ks] = mnext[4 * ks];   // in flight under the MFMAs
        }
        // the 4 rows of this lane group as whole rows (lane = 4 features), for the weighted
        // sum after the softmax: issued now so that their latency hides under the MFMAs
        float4_t mz[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = r0 + 4 * l4 + rr;
            mz[rr] = *reinterpret_cast<const float4_t *>(M + (row < R ? row : R - 1) * 64 + 4 * l15);
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const float bfrag = WREG ? wf[WREG ? ks : 0][t] : smem[(4 * ks + l4) * WLD + 16 * t + l15];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ks], bfrag, acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) afrag[ks] = anext[ks];
        float sc[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < TA; ++t) s += fast_tanh(acc[t][reg] + bcol[t]) * ucol[t];
            s = han_row16_sum(s);
            sc[reg] = s;     // every lane of the group holds the score of row 4*l4 + reg
        }
        if constexpr (P <= 4) {
#pragma unroll
            for (int k = 0; k < 4 / P; ++k) {       // the 4/P nodes of this lane group
                const int64_t row = r0 + 4 * l4 + k * P;
                if (row < R) {
                    float mx = sc[k * P];
#pragma unroll
                    for (int p = 1; p < P; ++p) mx = fmaxf(mx, sc[k * P + p]);
                    float e[P], den = 0.f;
#pragma unroll
                    for (int p = 0; p < P; ++p) { e[p] = __expf(sc[k * P + p] - mx); den += e[p]; }
                    const float inv = 1.f / den;
                    float4_t z = {0.f, 0.f, 0.f, 0.f};
                    float mine = 0.f;
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float bp = e[p] * inv;
#pragma unroll
                        for (int c = 0; c < 4; ++c) z[c] += bp * mz[k * P + p][c];
                        mine = (l15 == p) ? bp : mine;
                    }
                    const int64_t node = row / P;
                    *reinterpret_cast<float4_t *>(Z + node * 64 + 4 * l15) = z;
                    if (l15 < P) beta[node * P + l15] = mine;
                }
            }
        } else {
            // P = 8 / 16: the node's rows sit in P/4 lane groups; every lane takes part in the
            // cross-group exchanges (nodes are aligned, so a valid node never mixes with padding)
            float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
            mx = node_xmax<P>(mx);
            float e[4], den = 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { e[reg] = __expf(sc[reg] - mx); den += e[reg]; }
            den = node_xsum<P>(den);
            const float inv = 1.f / den;
            float4_t z = {0.f, 0.f, 0.f, 0.f};
            float mine = 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float bp = e[reg] * inv;
#pragma unroll
                for (int c = 0; c < 4; ++c) z[c] += bp * mz[reg][c];
                mine = (l15 == reg) ? bp : mine;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) z[c] = node_xsum<P>(z[c]);
            const int64_t row = r0 + 4 * l4;
            if (row < R) {
                if (l4 % (P / 4) == 0) *reinterpret_cast<float4_t *>(Z + (row / P) * 64 + 4 * l15) = z;
                if (l15 < 4) beta[row + l15] = mine;      // beta is (N,P) flat == row index
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The wave-local forward on the bf16 matrix pipe with fp32-class accuracy (the "bf16 x 6" scheme of
// project.hip): M rows and Womega are split EXACTLY into three bf16 terms (truncation), six products
// per K = 32 step replace sixteen v_mfma_f32_16x16x4_f32 per 64-deep row -- 12 MFMAs of 16 cycles
// instead of 16 of 32 per column tile.  Womega is split once per block into LDS, transposed
// ([term][a][k], 144-B rows) so that a lane's 8 k-values are one 16-B read; a lane loads ITS 16 k-values
// of its row as four 16-byte loads (one tile ahead) and splits them in registers.  Everything after
// the contraction (tanh, score dot, per-node softmax, weighted sum) is the fp32 kernel's.
// ---------------------------------------------------------------------------------------------
typedef __bf16 sa_bf16x8 __attribute__((ext_vector_type(8)));
typedef int sa_i32x4 __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
constexpr int SA_WLDB = 160;     // bytes per LDS row of the transposed, split Womega: 64 bf16 + 32 B -- with rows of
                                 // 160 B a ds_read_b128 lane group (tools/lds_banks.py) lands on 64 different banks

__device__ __forceinline__ void sa_split(float x, uint32_t &h, uint32_t &m, uint32_t &l) {
    h = __float_as_uint(x) & 0xFFFF0000u;             // x == h + m + l exactly (8 + 8 + 8 significand bits)
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));
}
__device__ __forceinline__ uint32_t sa_pack(uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }
__device__ __forceinline__ f32x4 sa_mfma(const sa_i32x4 &a, const sa_i32x4 &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sa_bf16x8, a), __builtin_bit_cast(sa_bf16x8, b),
                                                   c, 0, 0, 0);
}
// 8 consecutive floats -> the three packed bf16x8 fragments.  Two values at a time: the two subtractions of the
// split are packed (v_pk_add_f32), 4.5 instead of 5.5 vector instructions per value.
__device__ __forceinline__ void sa_split8(const float (&v)[8], sa_i32x4 &fh, sa_i32x4 &fm, sa_i32x4 &fl) {
    typedef uint32_t sa_u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2_t x = {v[2 * q], v[2 * q + 1]};
        const sa_u32x2 hb = __builtin_bit_cast(sa_u32x2, x) & 0xFFFF0000u;
        const float2_t r1 = x - __builtin_bit_cast(float2_t, hb);
        const sa_u32x2 mb = __builtin_bit_cast(sa_u32x2, r1) & 0xFFFF0000u;
        const sa_u32x2 lb = __builtin_bit_cast(sa_u32x2, r1 - __builtin_bit_cast(float2_t, mb));
        fh[q] = (int)sa_pack(hb[0], hb[1]);
        fm[q] = (int)sa_pack(mb[0], mb[1]);
        fl[q] = (int)sa_pack(lb[0], lb[1]);
    }
}

template <int CA, int P>
__global__ __launch_bounds__(256) void sem_attn_fwd_wave_b6_kernel(const float *__restrict__ M, const float *Wg,
                                                                   const float *bw, const float *uw, float *Z,
                                                                   float *beta, int64_t N) {
    constexpr int A = 64 * CA;
    constexpr int TA = A / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [3][A][128 B], XOR-swizzled: the 16-byte chunk g of row a sits at chunk g ^ ((a >> 1) & 7).  Two 128-B rows span
    // the 64 banks, so the 16 lanes that read chunk g of 16 consecutive rows (8 of each parity) hit 64 different banks
    // WITHOUT padding the rows to 144 B -- 48 KB instead of 54 KB at A = 128: three blocks per CU instead of two (round 3)
    constexpr int FWLDB = 128;
    unsigned char *Wt = reinterpret_cast<unsigned char *>(smem);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // Womega (64 x A, row-major [k][a]) -> split, transposed: item (a, g) = 8 k-values 8g..8g+7 of column a
    for (int it = threadIdx.x; it < A * 8; it += 256) {
        const int acol = it % A, g = it / A;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Wg[(8 * g + j) * A + acol];
        sa_i32x4 fh, fm, fl;
        sa_split8(v, fh, fm, fl);
        unsigned char *dst = Wt + acol * FWLDB + ((g ^ ((acol >> 1) & 7)) * 16);
        *reinterpret_cast<sa_i32x4 *>(dst) = fh;
        *reinterpret_cast<sa_i32x4 *>(dst + A * FWLDB) = fm;
        *reinterpret_cast<sa_i32x4 *>(dst + 2 * A * FWLDB) = fl;
    }
    __syncthreads();
    float bcol[TA], ucol[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        bcol[t] = bw[16 * t + l15];
        ucol[t] = uw[16 * t + l15];
    }
    const int64_t R = N * P;
    const int64_t ntiles = (R + 15) / 16;
    const int64_t tstride = (int64_t)gridDim.x * 4;
    // this lane's 16 k-values of its row: k = 32 s + 8 l4 + j  (s = 0, 1; j < 8): four 16-byte loads
    float4_t araw[4];
    {
        const int64_t t0 = (int64_t)blockIdx.x * 4 + w;
        const int64_t ra = t0 * 16 + l15 < R ? t0 * 16 + l15 : R - 1;
        const float *mr = M + ra * 64 + 8 * l4;
        araw[0] = *reinterpret_cast<const float4_t *>(mr);
        araw[1] = *reinterpret_cast<const float4_t *>(mr + 4);
        araw[2] = *reinterpret_cast<const float4_t *>(mr + 32);
        araw[3] = *reinterpret_cast<const float4_t *>(mr + 36);
    }
    for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += tstride) {
        const int64_t r0 = tile * 16;
        f32x4 acc[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sa_i32x4 af[2][3];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const float v[8] = {araw[2 * s2][0], araw[2 * s2][1], araw[2 * s2][2], araw[2 * s2][3],
                                araw[2 * s2 + 1][0], araw[2 * s2 + 1][1], araw[2 * s2 + 1][2], araw[2 * s2 + 1][3]};
            sa_split8(v, af[s2][0], af[s2][1], af[s2][2]);
        }
        {
            const int64_t rn = (tile + tstride) * 16 + l15;
            const float *mr = M + (rn < R ? rn : R - 1) * 64 + 8 * l4;      // next tile, in flight under the MFMAs
            araw[0] = *reinterpret_cast<const float4_t *>(mr);
            araw[1] = *reinterpret_cast<const float4_t *>(mr + 4);
            araw[2] = *reinterpret_cast<const float4_t *>(mr + 32);
            araw[3] = *reinterpret_cast<const float4_t *>(mr + 36);
        }
        float4_t mz[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = r0 + 4 * l4 + rr;
            mz[rr] = *reinterpret_cast<const float4_t *>(M + (row < R ? row : R - 1) * 64 + 4 * l15);
        }
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            f32x4 c = acc[t];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const unsigned char *wb = Wt + (16 * t + l15) * FWLDB + (((4 * s2 + l4) ^ ((l15 >> 1) & 7)) * 16);
                const sa_i32x4 bh = *reinterpret_cast<const sa_i32x4 *>(wb);
                const sa_i32x4 bm = *reinterpret_cast<const sa_i32x4 *>(wb + A * FWLDB);
                const sa_i32x4 bl = *reinterpret_cast<const sa_i32x4 *>(wb + 2 * A * FWLDB);
                c = sa_mfma(af[s2][1], bm, c);      // small terms first
                c = sa_mfma(af[s2][2], bh, c);
                c = sa_mfma(af[s2][0], bl, c);
                c = sa_mfma(af[s2][1], bh, c);
                c = sa_mfma(af[s2][0], bm, c);
                c = sa_mfma(af[s2][0], bh, c);
            }
            acc[t] = c;
            __builtin_amdgcn_sched_barrier(0);      // keep one column tile's 6 fragment reads in flight, not all 48
        }
        float sc[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < TA; ++t) s += fast_tanh(acc[t][reg] + bcol[t]) * ucol[t];
            s = han_row16_sum(s);
            sc[reg] = s;
        }
        if constexpr (P <= 4) {
#pragma unroll
            for (int k = 0; k < 4 / P; ++k) {
                const int64_t row = r0 + 4 * l4 + k * P;
                if (row < R) {
                    float mx = sc[k * P];
#pragma unroll
                    for (int p = 1; p < P; ++p) mx = fmaxf(mx, sc[k * P + p]);
                    float e[P], den = 0.f;
#pragma unroll
                    for (int p = 0; p < P; ++p) { e[p] = __expf(sc[k * P + p] - mx); den += e[p]; }
                    const float inv = 1.f / den;
                    float4_t z = {0.f, 0.f, 0.f, 0.f};
                    float mine = 0.f;
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float bp = e[p] * inv;
#pragma unroll
                        for (int c = 0; c < 4; ++c) z[c] += bp * mz[k * P + p][c];
                        mine = (l15 == p) ? bp : mine;
                    }
                    const int64_t node = row / P;
                    *reinterpret_cast<float4_t *>(Z + node * 64 + 4 * l15) = z;
                    if (l15 < P) beta[node * P + l15] = mine;
                }
            }
        } else {
            float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
            mx = node_xmax<P>(mx);
            float e[4], den = 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { e[reg] = __expf(sc[reg] - mx); den += e[reg]; }
            den = node_xsum<P>(den);
            const float inv = 1.f / den;
            float4_t z = {0.f, 0.f, 0.f, 0.f};
            float mine = 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float bp = e[reg] * inv;
#pragma unroll
                for (int c = 0; c < 4; ++c) z[c] += bp * mz[reg][c];
                mine = (l15 == reg) ? bp : mine;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) z[c] = node_xsum<P>(z[c]);
            const int64_t row = r0 + 4 * l4;
            if (row < R) {
                if (l4 % (P / 4) == 0) *reinterpret_cast<float4_t *>(Z + (row / P) * 64 + 4 * l15) = z;
                if (l15 < 4) beta[row + l15] = mine;
            }
        }
    }
}

// slab row per block: [64*A] dW | [A] db | [A] du
template <int CA>
__global__ __launch_bounds__(256) void sem_attn_bwd_kernel(const float *__restrict__ M, const float *Wg,
                                                           const float *bw, const float *uw,
                                                           const float *beta, const float *dZ, float *dM,
                                                           float *slab, int64_t N, int P) {
    constexpr int A = 64 * CA;
    constexpr int TA = A / 16;
    constexpr int WLD1 = A + 16;   // G1 B-operand reads: lanes step a, lane groups step f
    constexpr int WLD2 = A + 2;    // G2 reads: lanes step f / r, lane groups step a
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *W1 = smem;                         // [64][WLD1]
    float *W2 = W1 + 64 * WLD1;               // [64][WLD2]
    float *dp = W2 + 64 * WLD2;               // [4 waves][16][WLD2]
    float *dsb = dp + 4 * 16 * WLD2;          // [2][ROWS]  d s_r
    float *btb = dsb + 2 * ROWS;              // [2][ROWS]  beta_r
    for (int i = threadIdx.x; i < 64 * A; i += 256) {
        const float v = Wg[i];
        W1[(i / A) * WLD1 + (i % A)] = v;
        W2[(i / A) * WLD2 + (i % A)] = v;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    float bcol[TA], ucol[TA], du[TA], db[TA];
    f32x4 dW[4][TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        bcol[t] = bw[16 * t + l15];
        ucol[t] = uw[16 * t + l15];
        du[t] = 0.f;
        db[t] = 0.f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) dW[ft][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float *mydp = dp + w * 16 * WLD2;
    __syncthreads();
    const int NB = ROWS / P;
    const int64_t nchunks = (N + NB - 1) / NB;
    int buf = 0;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x, buf ^= 1) {
        const int64_t node0 = ch * NB;
        const int nodes = (int)((N - node0) < NB ? (N - node0) : NB);
        const int rows = nodes * P;
        const int64_t row0 = node0 * P;
        float *dsr = dsb + buf * ROWS, *btr = btb + buf * ROWS;
        // ---- P0: d beta, d s per node (lane = feature)
        if (threadIdx.x < ROWS && threadIdx.x >= rows) {
            dsr[threadIdx.x] = 0.f;
            btr[threadIdx.x] = 0.f;
        }
        for (int nd = w; nd < nodes; nd += 4) {
            const int64_t n = node0 + nd;
            const float dz = dZ[n * 64 + lane];
            const float breg = lane < P ? beta[n * P + lane] : 0.f;
            float dbreg = 0.f;
            for (int p = 0; p < P; ++p) {
                const float d = han_wave_sum(dz * M[(n * P + p) * 64 + lane]);
                if (lane == p) dbreg = d;
            }
            const float S = han_wave_sum(breg * dbreg);
            if (lane < P) {
                dsr[nd * P + lane] = breg * (dbreg - S);
                btr[nd * P + lane] = breg;
            }
        }
        __syncthreads();
        // ---- G1: pre = M_tile . Womega
        f32x4 acc[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const int lr = 16 * w + l15;
            const float *mrow = M + (row0 + (lr < rows ? lr : rows - 1)) * 64 + l4;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const float a = mrow[4 * ks];
                const float *wr = W1 + (4 * ks + l4) * WLD1 + l15;
#pragma unroll
                for (int t = 0; t < TA; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wr[16 * t], acc[t], 0, 0, 0);
            }
        }
        // ---- dpre in the accumulator layout (row r = 4*l4 + reg, col a = 16t + l15)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float ds = dsr[16 * w + 4 * l4 + reg];
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const float v = fast_tanh(acc[t][reg] + bcol[t]);
                const float d = ds * ucol[t] * (1.f - v * v);
                du[t] += ds * v;
                db[t] += d;
                acc[t][reg] = d;
                mydp[(4 * l4 + reg) * WLD2 + 16 * t + l15] = d;
            }
        }
        // ---- G3: dW += M_tile^T . dpre   (k-step `reg` holds rows 4g + reg, g = lane group)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int lr = 16 * w + 4 * l4 + reg;
            const float *mrow = M + (row0 + (lr < rows ? lr : rows - 1)) * 64 + l15;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const float a = mrow[16 * ft];
#pragma unroll
                for (int t = 0; t < TA; ++t)
                    dW[ft][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc[t][reg], dW[ft][t], 0, 0, 0);
            }
        }
        // ---- G2: dMx = dpre . Womega^T   (A operand from the wave's LDS tile)
        f32x4 acc2[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) acc2[ft] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int ks = 0; ks < A / 4; ++ks) {
            const float a = mydp[l15 * WLD2 + 4 * ks + l4];
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const float b = W2[(16 * ft + l15) * WLD2 + 4 * ks + l4];
                acc2[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2[ft], 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int lr = 16 * w + 4 * l4 + reg;
            if (lr < rows) {
                const int64_t n = node0 + lr / P;
                const float bt = btr[lr];
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    const int f = 16 * ft + l15;
                    dM[(row0 + lr) * 64 + f] = acc2[ft][reg] + bt * dZ[n * 64 + f];
                }
            }
        }
    }
    // ---- parameter gradients: lane groups -> waves (LDS) -> slab row
#pragma unroll
    for (int t = 0; t < TA; ++t) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            du[t] += __shfl_xor(du[t], o, 64);
            db[t] += __shfl_xor(db[t], o, 64);
        }
    }
    __syncthreads();
    float *red = smem;   // [64*A] dW | [A] db | [A] du  (fits inside W1 + W2)
    for (int ww = 0; ww < 4; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                for (int t = 0; t < TA; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int idx = (16 * ft + 4 * l4 + reg) * A + 16 * t + l15;
                        red[idx] = (ww == 0 ? 0.f : red[idx]) + dW[ft][t][reg];
                    }
            if (l4 == 0) {
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    const int idx = 64 * A + 16 * t + l15;
                    red[idx] = (ww == 0 ? 0.f : red[idx]) + db[t];
                    red[idx + A] = (ww == 0 ? 0.f : red[idx + A]) + du[t];
                }
            }
        }
        __syncthreads();
    }
    float *out = slab + (int64_t)blockIdx.x * (64 * A + 2 * A);
    for (int i = threadIdx.x; i < 64 * A + 2 * A; i += 256) out[i] = red[i];
}

// Backward for P in {1,2,4,8,16}: as in the forward, the P rows of a node share one
// 16-lane group of the accumulator layout (P/4 groups for P = 8, 16: one or two cross-group
// exchanges), so d beta / d s are formed inside the
// group (lane = 4 features of a row) and the whole tile flow G1 -> dpre -> G3 -> G2
// is wave-local: no block barrier in the main loop.
template <int CA, int P>
__global__ __launch_bounds__(256) void sem_attn_bwd_wave_kernel(const float *__restrict__ M, const float *Wg,
                                                                const float *bw, const float *uw,
                                                                const float *beta, const float *dZ, float *dM,
                                                                float *slab, int64_t N) {
    constexpr int A = 64 * CA;
    constexpr int TA = A / 16;
    constexpr int WLD1 = A + 16;
    constexpr int WLD2 = A + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *W1 = smem;                         // [64][WLD1]
    float *W2 = W1 + 64 * WLD1;               // [64][WLD2]
    float *dp = W2 + 64 * WLD2;               // [4 waves][16][WLD2]
    for (int i = threadIdx.x; i < 64 * A; i += 256) {
        const float v = Wg[i];
        W1[(i / A) * WLD1 + (i % A)] = v;
        W2[(i / A) * WLD2 + (i % A)] = v;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    float bcol[TA], ucol[TA], du[TA], db[TA];
    f32x4 dW[4][TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        bcol[t] = bw[16 * t + l15];
        ucol[t] = uw[16 * t + l15];
        du[t] = 0.f;
        db[t] = 0.f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) dW[ft][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float *mydp = dp + w * 16 * WLD2;
    __syncthreads();
    const int64_t R = N * P;
    const int64_t ntiles = (R + 15) / 16;
    // The rows of a tile come from HBM and with one wave per SIMD nothing else hides that
    // latency, so the row-local inputs (M rows, dZ rows, beta) are fetched ONE TILE AHEAD into
    // registers -- which also pulls the tile into L2 for the strided fragment loads of G1 / G3
    // (3.47 -> 3.20 ms at SYN-1M).
    const int64_t tstride = (int64_t)gridDim.x * 4;
    float4_t mv_n[4], dz_n[4];
    float bt_n[4];
    auto fetch_rows = [&](int64_t tile) {
        const int64_t g0n = tile * 16 + 4 * l4;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = g0n + rr;
            const bool ok = row < R;
            const int64_t rc = ok ? row : R - 1;
            mv_n[rr] = *reinterpret_cast<const float4_t *>(M + rc * 64 + 4 * l15);
            dz_n[rr] = *reinterpret_cast<const float4_t *>(dZ + (rc / P) * 64 + 4 * l15);
            bt_n[rr] = ok ? beta[rc] : 0.f;        // beta is (N,P) flat == row index
        }
    };
    fetch_rows((int64_t)blockIdx.x * 4 + w);
    float afrag[16];
    {
        const int64_t t0 = (int64_t)blockIdx.x * 4 + w;
        const int64_t ra = t0 * 16 + l15 < R ? t0 * 16 + l15 : R - 1;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) afrag[ks] = M[ra * 64 + l4 + 4 * ks];
    }
    for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += tstride) {
        const int64_t r0 = tile * 16;
        const int64_t g0 = r0 + 4 * l4;           // first row of this lane group
        // ---- d beta, d s inside the lane group (lane = features 4*l15 .. 4*l15+3)
        float ds[4], bt[4];
        {
            float dbt[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float4_t mv = mv_n[rr], dz = dz_n[rr];
                float d = mv[0] * dz[0] + mv[1] * dz[1] + mv[2] * dz[2] + mv[3] * dz[3];
                d = han_row16_sum(d);
                dbt[rr] = d;
                bt[rr] = bt_n[rr];
            }
            fetch_rows(tile + tstride);      // next tile: in flight under this tile's MFMAs
            if constexpr (P <= 4) {
#pragma unroll
                for (int k = 0; k < 4 / P; ++k) {
                    float S = 0.f;
#pragma unroll
                    for (int p2 = 0; p2 < P; ++p2) S += bt[k * P + p2] * dbt[k * P + p2];
#pragma unroll
                    for (int p2 = 0; p2 < P; ++p2) ds[k * P + p2] = bt[k * P + p2] * (dbt[k * P + p2] - S);
                }
            } else {     // P = 8 / 16: the node's rows span P/4 lane groups
                float S = 0.f;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) S += bt[rr] * dbt[rr];
                S = node_xsum<P>(S);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) ds[rr] = bt[rr] * (dbt[rr] - S);
            }
        }
        // ---- G1: pre = M_tile . Womega
        f32x4 acc[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float dzt[4][4];      // dZ of this tile's rows in the accumulator layout (L2 hits, used by G2)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t rg = g0 + reg < R ? g0 + reg : R - 1;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) dzt[reg][ft] = dZ[(rg / P) * 64 + 16 * ft + l15];
        }
        {
            float anext[16];      // G1's A fragments of the NEXT tile (as in the forward kernel)
            {
                const int64_t rn = (tile + tstride) * 16 + l15;
                const float *mnext = M + (rn < R ? rn : R - 1) * 64 + l4;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) anext[ks] = mnext[4 * ks];
            }
            // G3's A operand: this tile's rows, feature-major per lane (L2 hits)
            float g3a[4][4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t rg = g0 + reg < R ? g0 + reg : R - 1;     // dpre of a padding row is 0
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) g3a[reg][ft] = M[rg * 64 + l15 + 16 * ft];
            }
            // Column tile by column tile: G1(t) -> dpre(t) -> G3(t).  The VALU work of dpre(t)
            // (tanh, products, LDS store) has no dependence on the MFMAs of G1(t+1), so the
            // scheduler can run it in their shadow instead of after all of G1.
#pragma unroll
            for (int t = 0; t < TA; ++t) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        afrag[ks], W1[(4 * ks + l4) * WLD1 + l15 + 16 * t], acc[t], 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float v = fast_tanh(acc[t][reg] + bcol[t]);
                    const float d = ds[reg] * ucol[t] * (1.f - v * v);
                    du[t] += ds[reg] * v;
                    db[t] += d;
                    acc[t][reg] = d;
                    mydp[(4 * l4 + reg) * WLD2 + 16 * t + l15] = d;
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int ft = 0; ft < 4; ++ft)
                        dW[ft][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(g3a[reg][ft], acc[t][reg], dW[ft][t], 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) afrag[ks] = anext[ks];
        }
        // ---- G2: dMx = dpre . Womega^T   (A operand from the wave's LDS tile)
        // the accumulators start at beta * dZ (the direct term of dM); its loads were issued
        // before G1, so the epilogue is stores only
        f32x4 acc2[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc2[ft][reg] = bt[reg] * dzt[reg][ft];
#pragma unroll 8
        for (int ks = 0; ks < A / 4; ++ks) {
            const float a = mydp[l15 * WLD2 + 4 * ks + l4];
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const float b = W2[(16 * ft + l15) * WLD2 + 4 * ks + l4];
                acc2[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2[ft], 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t row = g0 + reg;
            if (row < R) {
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) dM[row * 64 + 16 * ft + l15] = acc2[ft][reg];
            }
        }
    }
    // ---- parameter gradients: lane groups -> waves (LDS) -> slab row
#pragma unroll
    for (int t = 0; t < TA; ++t) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            du[t] += __shfl_xor(du[t], o, 64);
            db[t] += __shfl_xor(db[t], o, 64);
        }
    }
    __syncthreads();
    float *red = smem;   // [64*A] dW | [A] db | [A] du  (fits inside W1)
    for (int ww = 0; ww < 4; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                for (int t = 0; t < TA; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int idx = (16 * ft + 4 * l4 + reg) * A + 16 * t + l15;
                        red[idx] = (ww == 0 ? 0.f : red[idx]) + dW[ft][t][reg];
                    }
            if (l4 == 0) {
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    const int idx = 64 * A + 16 * t + l15;
                    red[idx] = (ww == 0 ? 0.f : red[idx]) + db[t];
                    red[idx + A] = (ww == 0 ? 0.f : red[idx + A]) + du[t];
                }
            }
        }
        __syncthreads();
    }
    float *out = slab + (int64_t)blockIdx.x * (64 * A + 2 * A);
    for (int i = threadIdx.x; i < 64 * A + 2 * A; i += 256) out[i] = red[i];
}

// First attention column of the eight a lane group l4 contributes to step s of G2's reduction (see the kernel's header).
template <int CA>
__device__ __forceinline__ int g2_col(int s, int l4) {
    if constexpr (CA == 2) return 64 * (l4 & 1) + 32 * (l4 >> 1) + 8 * s;
    else return 32 * s + 8 * l4;
}

// The wave-local backward with G1 (recompute of pre) and G2 (dM = dpre . Womega^T) on the bf16 matrix pipe
// (exact 3-way split, fp32-class accuracy: see sem_attn_fwd_wave_b6_kernel).  G3 (dWomega += M^T dpre) reduces over
// ROWS, 16 per tile -- half a K = 32 step: the wave therefore works on two of its tiles per loop pass and runs G3
// once for both, also on the bf16 pipe (G3B; k-slot (l4, j) = tile j / 4, row 4 l4 + j % 4 -- exactly the four dpre
// values per tile a lane's G1 accumulators hold and the four rows per tile whose features it loaded).  One wave
// issues an fp32 16x16x4 MFMA every ~51 cycles but a bf16 16x16x32 every ~27
// (profiles/r04_ubench_mfma_valu_overlap.jsonl): 96 bf16 instead of 128 fp32 MFMAs per tile, for ~260 more vector
// instructions of splitting.  G3B = false keeps G3 on the fp32 pipe (tile by tile; measurements).  Womega is split
// twice into LDS at block start: transposed [a][k] for G1, [f][a] for G2.
//
// Which feature a lane index stands for is chosen so that no operand is loaded twice: in G3's A operand and in
// G2's output, index i of the 16 x 16 tile number ft is feature 4 i + ft -- the four features a lane already holds
// of each of its rows (the float4 row loads of M and dZ) are then its G3 operands and the start values of its G2
// accumulators, and a row of dM leaves as one 16-byte store per lane.  G2's reduction index (the attention
// column) is ordered so that the 16-byte reads of the dpre tile and of the split Womega are bank-conflict free
// (CA = 2: k-slot (l4, j) of step s is column 64 (l4 & 1) + 32 (l4 >> 1) + 8 s + j).
template <int CA, int P, bool G3B>
__global__ __launch_bounds__(256) void sem_attn_bwd_wave_b6_kernel(const float *__restrict__ M, const float *Wg,
                                                                const float *bw, const float *uw,
                                                                const float *beta, const float *dZ, float *dM,
                                                                float *slab, int64_t N) {
    constexpr int A = 64 * CA;
    constexpr int TA = A / 16;
    constexpr int WLD2 = A + 4;                 // dpre tile rows: 16-B aligned for the 8-float fragment reads
    constexpr int W2LDB = A * 2 + 32;           // bytes per row of the split Womega [f][a] (G2's B operand)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char *Wt = reinterpret_cast<unsigned char *>(smem);        // G1 B operand: [3][A][144 B]   (transposed [a][k])
    unsigned char *W2s = Wt + 3 * A * SA_WLDB;                          // G2 B operand: [3][64 f][W2LDB] (as stored [f][a])
    float *dp = reinterpret_cast<float *>(W2s + 3 * 64 * W2LDB);        // [4 waves][16][WLD2] fp32
    // b_omega | u_omega: held in registers up to P = 4; the P = 8 / 16 variants (node sums across lane groups) would
    // spill five registers, and read the two values of a column tile from LDS instead (3 % slower at P = 4: measured)
    constexpr bool BU_LDS = G3B && P >= 8;
    float *bus = dp + 4 * 16 * WLD2;
    if (BU_LDS)
        for (int it = threadIdx.x; it < 2 * A; it += 256) bus[it] = it < A ? bw[it] : uw[it - A];
    for (int it = threadIdx.x; it < A * 8; it += 256) {                 // (a, g): k-values 8g..8g+7 of column a
        const int acol = it % A, g = it / A;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Wg[(8 * g + j) * A + acol];
        sa_i32x4 fh, fm, fl;
        sa_split8(v, fh, fm, fl);
        unsigned char *dst = Wt + acol * SA_WLDB + g * 16;
        *reinterpret_cast<sa_i32x4 *>(dst) = fh;
        *reinterpret_cast<sa_i32x4 *>(dst + A * SA_WLDB) = fm;
        *reinterpret_cast<sa_i32x4 *>(dst + 2 * A * SA_WLDB) = fl;
    }
    for (int it = threadIdx.x; it < 64 * (A / 8); it += 256) {          // (row, g): the g-th 16-byte piece of an LDS row
        const int lr = it / (A / 8), g = it % (A / 8);
        const int f = 4 * (lr & 15) + (lr >> 4);                        // row 16 ft + i holds feature 4 i + ft
        const int a0 = g2_col<CA>(g >> 2, g & 3);                       // piece g = step g / 4, lane group g % 4
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Wg[f * A + a0 + j];
        sa_i32x4 fh, fm, fl;
        sa_split8(v, fh, fm, fl);
        unsigned char *dst = W2s + lr * W2LDB + g * 16;
        *reinterpret_cast<sa_i32x4 *>(dst) = fh;
        *reinterpret_cast<sa_i32x4 *>(dst + 64 * W2LDB) = fm;
        *reinterpret_cast<sa_i32x4 *>(dst + 2 * 64 * W2LDB) = fl;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    float bcol[BU_LDS ? 1 : TA], ucol[BU_LDS ? 1 : TA], du[TA], db[TA];
    f32x4 dW[4][TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        if constexpr (!BU_LDS) {
            bcol[t] = bw[16 * t + l15];
            ucol[t] = uw[16 * t + l15];
        }
        du[t] = 0.f;
        db[t] = 0.f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) dW[ft][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float *mydp = dp + w * 16 * WLD2;
    __syncthreads();
    const int64_t R = N * P;
    const int64_t ntiles = (R + 15) / 16;
    // The rows of a tile come from HBM and with one wave per SIMD nothing else hides that
    // latency, so the row-local inputs (M rows, dZ rows, beta) are fetched ONE TILE AHEAD into
    // registers -- which also pulls the tile into L2 for the strided fragment loads of G1 / G3
    // (3.47 -> 3.20 ms at SYN-1M).
    const int64_t tstride = (int64_t)gridDim.x * 4;
    float4_t mv_n[4], dz_n[4];
    float bt_n[4];
    auto fetch_rows = [&](int64_t tile) {
        const int64_t g0n = tile * 16 + 4 * l4;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = g0n + rr;
            const bool ok = row < R;
            const int64_t rc = ok ? row : R - 1;
            mv_n[rr] = *reinterpret_cast<const float4_t *>(M + rc * 64 + 4 * l15);
            dz_n[rr] = *reinterpret_cast<const float4_t *>(dZ + (rc / P) * 64 + 4 * l15);
            bt_n[rr] = ok ? beta[rc] : 0.f;        // beta is (N,P) flat == row index
        }
    };
    fetch_rows((int64_t)blockIdx.x * 4 + w);
    float4_t araw[4];      // G1: this lane's 16 k-values of its row (k = 32 s + 8 l4 + j), one tile ahead
    {
        const int64_t t0 = (int64_t)blockIdx.x * 4 + w;
        const int64_t ra = t0 * 16 + l15 < R ? t0 * 16 + l15 : R - 1;
        const float *mr = M + ra * 64 + 8 * l4;
        araw[0] = *reinterpret_cast<const float4_t *>(mr);
        araw[1] = *reinterpret_cast<const float4_t *>(mr + 4);
        araw[2] = *reinterpret_cast<const float4_t *>(mr + 32);
        araw[3] = *reinterpret_cast<const float4_t *>(mr + 36);
    }
    for (int64_t tile0 = (int64_t)blockIdx.x * 4 + w; tile0 < ntiles; tile0 += 2 * tstride) {
      f32x4 dps[TA];            // G3B: dpre of the pass's first tile, kept until the second tile's column tile t is done
      float4_t mvs[4];          // G3B: the first tile's rows
      sa_i32x4 mf[4][3];        // G3B: G3's A fragments, M^T of both tiles (k-slot (l4, j) = tile j / 4, row 4 l4 + j % 4)
      // the second tile of the last pass may lie past the end: its rows are clamped, beta = 0 makes every one of
      // its contributions zero and its stores are masked
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t tile = tile0 + u * tstride;
        if (!G3B && tile >= ntiles) break;
        const int64_t r0 = tile * 16;
        const int64_t g0 = r0 + 4 * l4;           // first row of this lane group
        // ---- d beta, d s inside the lane group (lane = features 4*l15 .. 4*l15+3)
        float ds[4], bt[4];
        float4_t mvc[4], dzc[4];      // this tile's rows: features 4 l15 .. 4 l15 + 3 of rows g0 .. g0 + 3
        {
            float dbt[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float4_t mv = mv_n[rr], dz = dz_n[rr];
                mvc[rr] = mv;
                if (G3B && u == 0) mvs[rr] = mv;
                dzc[rr] = dz;
                float d = mv[0] * dz[0] + mv[1] * dz[1] + mv[2] * dz[2] + mv[3] * dz[3];
                d = han_row16_sum(d);
                dbt[rr] = d;
                bt[rr] = bt_n[rr];
            }
            fetch_rows(tile + tstride);      // next tile: in flight under this tile's MFMAs
            if constexpr (P <= 4) {
#pragma unroll
                for (int k = 0; k < 4 / P; ++k) {
                    float S = 0.f;
#pragma unroll
                    for (int p2 = 0; p2 < P; ++p2) S += bt[k * P + p2] * dbt[k * P + p2];
#pragma unroll
                    for (int p2 = 0; p2 < P; ++p2) ds[k * P + p2] = bt[k * P + p2] * (dbt[k * P + p2] - S);
                }
            } else {     // P = 8 / 16: the node's rows span P/4 lane groups
                float S = 0.f;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) S += bt[rr] * dbt[rr];
                S = node_xsum<P>(S);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) ds[rr] = bt[rr] * (dbt[rr] - S);
            }
        }
        if (G3B && u == 1) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const float v[8] = {mvs[0][ft], mvs[1][ft], mvs[2][ft], mvs[3][ft],
                                    mvc[0][ft], mvc[1][ft], mvc[2][ft], mvc[3][ft]};
                sa_split8(v, mf[ft][0], mf[ft][1], mf[ft][2]);
            }
        }
        // ---- G1: pre = M_tile . Womega
        f32x4 acc[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            sa_i32x4 af[2][3];     // G1's A fragments: exact 3-way bf16 split of this tile's rows
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float v[8] = {araw[2 * s2][0], araw[2 * s2][1], araw[2 * s2][2], araw[2 * s2][3],
                                    araw[2 * s2 + 1][0], araw[2 * s2 + 1][1], araw[2 * s2 + 1][2], araw[2 * s2 + 1][3]};
                sa_split8(v, af[s2][0], af[s2][1], af[s2][2]);
            }
            {
                const int64_t rn = (tile + tstride) * 16 + l15;      // next tile's rows, in flight under the MFMAs
                const float *mr = M + (rn < R ? rn : R - 1) * 64 + 8 * l4;
                araw[0] = *reinterpret_cast<const float4_t *>(mr);
                araw[1] = *reinterpret_cast<const float4_t *>(mr + 4);
                araw[2] = *reinterpret_cast<const float4_t *>(mr + 32);
                araw[3] = *reinterpret_cast<const float4_t *>(mr + 36);
            }
            // G3's A operand is mvc: index i of tile ft = feature 4 i + ft (dpre of a padding row is 0)
            // Column tile by column tile: G1(t) -> dpre(t) -> G3(t).  The VALU work of dpre(t)
            // (tanh, products, LDS store) has no dependence on the MFMAs of G1(t+1), so the
            // scheduler can run it in their shadow instead of after all of G1.
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                {
                    f32x4 c = acc[t];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const unsigned char *wb = Wt + (16 * t + l15) * SA_WLDB + (32 * s2 + 8 * l4) * 2;
                        const sa_i32x4 bh = *reinterpret_cast<const sa_i32x4 *>(wb);
                        const sa_i32x4 bm = *reinterpret_cast<const sa_i32x4 *>(wb + A * SA_WLDB);
                        const sa_i32x4 bl = *reinterpret_cast<const sa_i32x4 *>(wb + 2 * A * SA_WLDB);
                        c = sa_mfma(af[s2][1], bm, c);
                        c = sa_mfma(af[s2][2], bh, c);
                        c = sa_mfma(af[s2][0], bl, c);
                        c = sa_mfma(af[s2][1], bh, c);
                        c = sa_mfma(af[s2][0], bm, c);
                        c = sa_mfma(af[s2][0], bh, c);
                    }
                    acc[t] = c;
                }
                float bcol_t, ucol_t;
                if constexpr (BU_LDS) {
                    bcol_t = bus[16 * t + l15];
                    ucol_t = bus[A + 16 * t + l15];
                } else {
                    bcol_t = bcol[t];
                    ucol_t = ucol[t];
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float v = fast_tanh(acc[t][reg] + bcol_t);
                    const float d = ds[reg] * ucol_t * (1.f - v * v);
                    du[t] += ds[reg] * v;
                    db[t] += d;
                    acc[t][reg] = d;
                    mydp[(4 * l4 + reg) * WLD2 + 16 * t + l15] = d;
                }
                if constexpr (G3B) {
                    // G3 of both tiles' column tile t: dWomega[4 i + ft][16 t + n] += sum over the 32 rows of
                    // M[row][4 i + ft] dpre[row][16 t + n]
                    if (u == 0) {
                        dps[t] = acc[t];
                    } else {
                        const float v[8] = {dps[t][0], dps[t][1], dps[t][2], dps[t][3],
                                            acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
                        sa_i32x4 dh, dm, dl;
                        sa_split8(v, dh, dm, dl);
#pragma unroll
                        for (int ft = 0; ft < 4; ++ft) {
                            f32x4 c = dW[ft][t];
                            c = sa_mfma(mf[ft][1], dm, c);
                            c = sa_mfma(mf[ft][2], dh, c);
                            c = sa_mfma(mf[ft][0], dl, c);
                            c = sa_mfma(mf[ft][1], dh, c);
                            c = sa_mfma(mf[ft][0], dm, c);
                            c = sa_mfma(mf[ft][0], dh, c);
                            dW[ft][t] = c;
                        }
                    }
                } else {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                        for (int ft = 0; ft < 4; ++ft)
                            dW[ft][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(mvc[reg][ft], acc[t][reg], dW[ft][t], 0, 0, 0);
                }
            }
        }
        // ---- G2: dMx = dpre . Womega^T   (A operand from the wave's LDS tile)
        // the accumulators start at beta * dZ (the direct term of dM), so the epilogue is stores only
        f32x4 acc2[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc2[ft][reg] = bt[reg] * dzc[reg][ft];
#pragma unroll
        for (int s2 = 0; s2 < A / 32; ++s2) {       // K = 32 columns of the attention space per step
            const float4_t d0 = *reinterpret_cast<const float4_t *>(mydp + l15 * WLD2 + g2_col<CA>(s2, l4));
            const float4_t d1 = *reinterpret_cast<const float4_t *>(mydp + l15 * WLD2 + g2_col<CA>(s2, l4) + 4);
            const float v[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
            sa_i32x4 ah, am, al;
            sa_split8(v, ah, am, al);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const unsigned char *wb = W2s + (16 * ft + l15) * W2LDB + (4 * s2 + l4) * 16;
                const sa_i32x4 bh = *reinterpret_cast<const sa_i32x4 *>(wb);
                const sa_i32x4 bm = *reinterpret_cast<const sa_i32x4 *>(wb + 64 * W2LDB);
                const sa_i32x4 bl = *reinterpret_cast<const sa_i32x4 *>(wb + 2 * 64 * W2LDB);
                f32x4 c = acc2[ft];
                c = sa_mfma(am, bm, c);
                c = sa_mfma(al, bh, c);
                c = sa_mfma(ah, bl, c);
                c = sa_mfma(am, bh, c);
                c = sa_mfma(ah, bm, c);
                c = sa_mfma(ah, bh, c);
                acc2[ft] = c;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t row = g0 + reg;
            if (row < R)
                *reinterpret_cast<float4_t *>(dM + row * 64 + 4 * l15) =
                    (float4_t){acc2[0][reg], acc2[1][reg], acc2[2][reg], acc2[3][reg]};
        }
      }
    }
    // ---- parameter gradients: lane groups -> waves (LDS) -> slab row
#pragma unroll
    for (int t = 0; t < TA; ++t) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            du[t] += __shfl_xor(du[t], o, 64);
            db[t] += __shfl_xor(db[t], o, 64);
        }
    }
    __syncthreads();
    float *red = smem;   // [64*A] dW | [A] db | [A] du  (fits inside the split-Womega region)
    for (int ww = 0; ww < 4; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                for (int t = 0; t < TA; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int idx = (16 * l4 + 4 * reg + ft) * A + 16 * t + l15;     // feature 4 i + ft, i = 4 l4 + reg
                        red[idx] = (ww == 0 ? 0.f : red[idx]) + dW[ft][t][reg];
                    }
            if (l4 == 0) {
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    const int idx = 64 * A + 16 * t + l15;
                    red[idx] = (ww == 0 ? 0.f : red[idx]) + db[t];
                    red[idx + A] = (ww == 0 ? 0.f : red[idx + A]) + du[t];
                }
            }
        }
        __syncthreads();
    }
    float *out = slab + (int64_t)blockIdx.x * (64 * A + 2 * A);
    for (int i = threadIdx.x; i < 64 * A + 2 * A; i += 256) out[i] = red[i];
}

// The same backward at A = 128 with TWO waves per SIMD.  One wave cannot keep the matrix pipe busy (a wave issues
// an fp32 16x16x4 MFMA every ~51 cycles and a bf16 16x16x32 every ~27 where the pipe takes 32 / 16:
// profiles/r04_ubench_mfma_valu_overlap.jsonl), and the kernel above needs 390 registers -- 128 of them the dW
// accumulators of a 64 x 128 matrix -- so it runs one wave per SIMD.  Here two waves SHARE a tile of 16 rows: wave h
// of the pair owns attention columns 64 h .. 64 h + 63 for G1, dpre and G3 (64 dW accumulators) and output tiles
// ft = 2 h, 2 h + 1 for G2, whose A operand is the pair's complete dpre tile in LDS.  Eight waves (four pairs) per
// block share one split Womega; the two block barriers per tile are executed by every wave for the same number of
// tiles (a pair whose tile lies past the end works on clamped rows with beta = 0: all its contributions are zero
// and its stores are masked).  The row-local work (d beta, d s) and G1's A fragments are computed by both waves.
template <int P>
__global__ __launch_bounds__(512) void sem_attn_bwd_pair_b6_kernel(const float *__restrict__ M, const float *Wg,
                                                                const float *bw, const float *uw,
                                                                const float *beta, const float *dZ, float *dM,
                                                                float *slab, int64_t N) {
    constexpr int CA = 2;
    constexpr int A = 128;
    constexpr int TH = 4;                       // column tiles per wave
    constexpr int WLD2 = A + 4;
    constexpr int W2LDB = A * 2 + 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char *Wt = reinterpret_cast<unsigned char *>(smem);        // G1 B operand: [3][A][SA_WLDB]
    unsigned char *W2s = Wt + 3 * A * SA_WLDB;                          // G2 B operand: [3][64][W2LDB]
    float *dp = reinterpret_cast<float *>(W2s + 3 * 64 * W2LDB);        // [4 pairs][16][WLD2] fp32
    for (int it = threadIdx.x; it < A * 8; it += 512) {
        const int acol = it % A, g = it / A;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Wg[(8 * g + j) * A + acol];
        sa_i32x4 fh, fm, fl;
        sa_split8(v, fh, fm, fl);
        unsigned char *dst = Wt + acol * SA_WLDB + g * 16;
        *reinterpret_cast<sa_i32x4 *>(dst) = fh;
        *reinterpret_cast<sa_i32x4 *>(dst + A * SA_WLDB) = fm;
        *reinterpret_cast<sa_i32x4 *>(dst + 2 * A * SA_WLDB) = fl;
    }
    for (int it = threadIdx.x; it < 64 * (A / 8); it += 512) {
        const int lr = it / (A / 8), g = it % (A / 8);
        const int f = 4 * (lr & 15) + (lr >> 4);
        const int a0 = g2_col<CA>(g >> 2, g & 3);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Wg[f * A + a0 + j];
        sa_i32x4 fh, fm, fl;
        sa_split8(v, fh, fm, fl);
        unsigned char *dst = W2s + lr * W2LDB + g * 16;
        *reinterpret_cast<sa_i32x4 *>(dst) = fh;
        *reinterpret_cast<sa_i32x4 *>(dst + 64 * W2LDB) = fm;
        *reinterpret_cast<sa_i32x4 *>(dst + 2 * 64 * W2LDB) = fl;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pair = w >> 1;
    const int h = __builtin_amdgcn_readfirstlane(w & 1);
    const int l15 = lane & 15, l4 = lane >> 4;
    float bcol[TH], ucol[TH], du[TH], db[TH];
    f32x4 dW[4][TH];
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        bcol[t] = bw[64 * h + 16 * t + l15];
        ucol[t] = uw[64 * h + 16 * t + l15];
        du[t] = 0.f;
        db[t] = 0.f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) dW[ft][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float *mydp = dp + pair * 16 * WLD2;
    __syncthreads();
    const int64_t R = N * P;
    const int64_t ntiles = (R + 15) / 16;
    const int64_t tstride = (int64_t)gridDim.x * 4;
    float4_t mv_n[4], dz_n[4];
    float bt_n[4];
    // the next tile's rows: dZ and beta are requested at the top of a tile, the M rows (which stay G3's A operand
    // until G3 is done) after G3 -- by then G1's fragment loads of the same rows have pulled them into the caches
    auto fetch_dz = [&](int64_t tile) {
        const int64_t g0n = tile * 16 + 4 * l4;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = g0n + rr;
            const bool ok = row < R;
            const int64_t rc = ok ? row : R - 1;
            dz_n[rr] = *reinterpret_cast<const float4_t *>(dZ + (rc / P) * 64 + 4 * l15);
            bt_n[rr] = ok ? beta[rc] : 0.f;
        }
    };
    auto fetch_mv = [&](int64_t tile) {
        const int64_t g0n = tile * 16 + 4 * l4;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t rc = g0n + rr < R ? g0n + rr : R - 1;
            mv_n[rr] = *reinterpret_cast<const float4_t *>(M + rc * 64 + 4 * l15);
        }
    };
    float4_t araw[4];
    auto fetch_a = [&](int64_t tile) {
        const int64_t ra = tile * 16 + l15 < R ? tile * 16 + l15 : R - 1;
        const float *mr = M + ra * 64 + 8 * l4;
        araw[0] = *reinterpret_cast<const float4_t *>(mr);
        araw[1] = *reinterpret_cast<const float4_t *>(mr + 4);
        araw[2] = *reinterpret_cast<const float4_t *>(mr + 32);
        araw[3] = *reinterpret_cast<const float4_t *>(mr + 36);
    };
    fetch_dz((int64_t)blockIdx.x * 4 + pair);
    fetch_mv((int64_t)blockIdx.x * 4 + pair);
    fetch_a((int64_t)blockIdx.x * 4 + pair);
    // every wave of the block runs the same number of tiles: the barriers below are block-wide
    for (int64_t base = (int64_t)blockIdx.x * 4; base < ntiles; base += tstride) {
        const int64_t tile = base + pair;
        const int64_t r0 = tile * 16;
        const int64_t g0 = r0 + 4 * l4;
        float ds[4], bt[4];
        f32x4 acc2[2];            // G2's accumulators start at beta * dZ (the direct term of dM)
        {
            float dbt[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float4_t mv = mv_n[rr], dz = dz_n[rr];
                bt[rr] = bt_n[rr];
                acc2[0][rr] = bt[rr] * (h ? dz[2] : dz[0]);
                acc2[1][rr] = bt[rr] * (h ? dz[3] : dz[1]);
                float d = mv[0] * dz[0] + mv[1] * dz[1] + mv[2] * dz[2] + mv[3] * dz[3];
                d = han_row16_sum(d);
                dbt[rr] = d;
            }
            fetch_dz(tile + tstride);
            if constexpr (P <= 4) {
#pragma unroll
                for (int k = 0; k < 4 / P; ++k) {
                    float S = 0.f;
#pragma unroll
                    for (int p2 = 0; p2 < P; ++p2) S += bt[k * P + p2] * dbt[k * P + p2];
#pragma unroll
                    for (int p2 = 0; p2 < P; ++p2) ds[k * P + p2] = bt[k * P + p2] * (dbt[k * P + p2] - S);
                }
            } else {
                float S = 0.f;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) S += bt[rr] * dbt[rr];
                S = node_xsum<P>(S);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) ds[rr] = bt[rr] * (dbt[rr] - S);
            }
        }
        {
            sa_i32x4 af[2][3];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float v[8] = {araw[2 * s2][0], araw[2 * s2][1], araw[2 * s2][2], araw[2 * s2][3],
                                    araw[2 * s2 + 1][0], araw[2 * s2 + 1][1], araw[2 * s2 + 1][2], araw[2 * s2 + 1][3]};
                sa_split8(v, af[s2][0], af[s2][1], af[s2][2]);
            }
            fetch_a(tile + tstride);
#pragma unroll
            for (int t = 0; t < TH; ++t) {
                f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const unsigned char *wb = Wt + (64 * h + 16 * t + l15) * SA_WLDB + (32 * s2 + 8 * l4) * 2;
                    const sa_i32x4 bh = *reinterpret_cast<const sa_i32x4 *>(wb);
                    const sa_i32x4 bm = *reinterpret_cast<const sa_i32x4 *>(wb + A * SA_WLDB);
                    const sa_i32x4 bl = *reinterpret_cast<const sa_i32x4 *>(wb + 2 * A * SA_WLDB);
                    c = sa_mfma(af[s2][1], bm, c);
                    c = sa_mfma(af[s2][2], bh, c);
                    c = sa_mfma(af[s2][0], bl, c);
                    c = sa_mfma(af[s2][1], bh, c);
                    c = sa_mfma(af[s2][0], bm, c);
                    c = sa_mfma(af[s2][0], bh, c);
                }
                if (t == 0) __syncthreads();      // the pair's previous tile has been read by both of its waves
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float v = fast_tanh(c[reg] + bcol[t]);
                    const float d = ds[reg] * ucol[t] * (1.f - v * v);
                    du[t] += ds[reg] * v;
                    db[t] += d;
                    c[reg] = d;
                    mydp[(4 * l4 + reg) * WLD2 + 64 * h + 16 * t + l15] = d;
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int ft = 0; ft < 4; ++ft)
                        dW[ft][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(mv_n[reg][ft], c[reg], dW[ft][t], 0, 0, 0);
            }
        }
        fetch_mv(tile + tstride);
        __syncthreads();                          // both halves of the pair's dpre tile are in LDS
#pragma unroll
        for (int s2 = 0; s2 < A / 32; ++s2) {
            const float4_t d0 = *reinterpret_cast<const float4_t *>(mydp + l15 * WLD2 + g2_col<CA>(s2, l4));
            const float4_t d1 = *reinterpret_cast<const float4_t *>(mydp + l15 * WLD2 + g2_col<CA>(s2, l4) + 4);
            const float v[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
            sa_i32x4 ah, am, al;
            sa_split8(v, ah, am, al);
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2) {
                const unsigned char *wb = W2s + (16 * (2 * h + f2) + l15) * W2LDB + (4 * s2 + l4) * 16;
                const sa_i32x4 bh = *reinterpret_cast<const sa_i32x4 *>(wb);
                const sa_i32x4 bm = *reinterpret_cast<const sa_i32x4 *>(wb + 64 * W2LDB);
                const sa_i32x4 bl = *reinterpret_cast<const sa_i32x4 *>(wb + 2 * 64 * W2LDB);
                f32x4 c = acc2[f2];
                c = sa_mfma(am, bm, c);
                c = sa_mfma(al, bh, c);
                c = sa_mfma(ah, bl, c);
                c = sa_mfma(am, bh, c);
                c = sa_mfma(ah, bm, c);
                c = sa_mfma(ah, bh, c);
                acc2[f2] = c;
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t row = g0 + reg;
            if (row < R)
                *reinterpret_cast<float2_t *>(dM + row * 64 + 4 * l15 + 2 * h) = (float2_t){acc2[0][reg], acc2[1][reg]};
        }
    }
    // ---- parameter gradients: lane groups -> pairs (LDS) -> slab row
#pragma unroll
    for (int t = 0; t < TH; ++t) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            du[t] += __shfl_xor(du[t], o, 64);
            db[t] += __shfl_xor(db[t], o, 64);
        }
    }
    __syncthreads();
    float *red = smem;   // [64*A] dW | [A] db | [A] du  (fits inside the split-Womega region)
    for (int pp = 0; pp < 4; ++pp) {
        if (pair == pp) {     // the two waves of a pair own different columns
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                for (int t = 0; t < TH; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int idx = (16 * l4 + 4 * reg + ft) * A + 64 * h + 16 * t + l15;
                        red[idx] = (pp == 0 ? 0.f : red[idx]) + dW[ft][t][reg];
                    }
            if (l4 == 0) {
#pragma unroll
                for (int t = 0; t < TH; ++t) {
                    const int idx = 64 * A + 64 * h + 16 * t + l15;
                    red[idx] = (pp == 0 ? 0.f : red[idx]) + db[t];
                    red[idx + A] = (pp == 0 ? 0.f : red[idx + A]) + du[t];
                }
            }
        }
        __syncthreads();
    }
    float *out = slab + (int64_t)blockIdx.x * (64 * A + 2 * A);
    for (int i = threadIdx.x; i < 64 * A + 2 * A; i += 512) out[i] = red[i];
}

// ---------------------------------------------------------------------------------------------
// Wider embeddings: D = 128 (e.g. hid_units = [16] with 8 heads, models/gat.py:42-57 leaves both
// free).  The block-level kernels above with the feature width as a template parameter
// (D = 16*DT); the D = 64 kernels stay as they are.  The backward keeps D*AS/64 dW accumulators
// per lane, so at D = 128 it runs over the attention space in slices of AS = 64 columns
// (one launch per slice; slices after the first ADD their dM contribution).
// ---------------------------------------------------------------------------------------------
template <int CA, int DT>
__global__ __launch_bounds__(256) void sem_attn_fwd_gen_kernel(const float *__restrict__ M, const float *Wg,
                                                               const float *bw, const float *uw, float *Z,
                                                               float *beta, int64_t N, int P) {
    constexpr int D = 16 * DT;
    constexpr int NF = D / 64;       // features per lane in the lane = feature phases
    constexpr int A = 64 * CA;
    constexpr int TA = A / 16;
    constexpr int WLD = A + 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;                 // [D][WLD]
    float *sc = smem + D * WLD;       // [2][ROWS]
    for (int i = threadIdx.x; i < D * A; i += 256) Wl[(i / A) * WLD + (i % A)] = Wg[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    float bcol[TA], ucol[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        bcol[t] = bw[16 * t + l15];
        ucol[t] = uw[16 * t + l15];
    }
    __syncthreads();
    const int NB = ROWS / P;
    const int64_t nchunks = (N + NB - 1) / NB;
    int buf = 0;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x, buf ^= 1) {
        const int64_t node0 = ch * NB;
        const int nodes = (int)((N - node0) < NB ? (N - node0) : NB);
        const int rows = nodes * P;
        const int64_t row0 = node0 * P;
        {
            const int lr = 16 * w + l15;
            const float *mrow = M + (row0 + (lr < rows ? lr : rows - 1)) * D + l4;
            f32x4 acc[TA];
#pragma unroll
            for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int ks = 0; ks < D / 4; ++ks) {
                const float a = mrow[4 * ks];
                const float *wr = Wl + (4 * ks + l4) * WLD + l15;
#pragma unroll
                for (int t = 0; t < TA; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wr[16 * t], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < TA; ++t) s += fast_tanh(acc[t][reg] + bcol[t]) * ucol[t];
                s = han_row16_sum(s);
                if (l15 == 0) sc[buf * ROWS + 16 * w + 4 * l4 + reg] = s;
            }
        }
        __syncthreads();
        for (int nd = w; nd < nodes; nd += 4) {
            const int64_t n = node0 + nd;
            float zacc[NF], mrun = HAN_NEG_BIG, lrun = 0.f, sreg = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) zacc[f] = 0.f;
            for (int p = 0; p < P; ++p) {
                const float s = sc[buf * ROWS + nd * P + p];
                if (lane == p) sreg = s;
                const float mn = fmaxf(mrun, s);
                const float scl = __expf(mrun - mn), pe = __expf(s - mn);
                lrun = lrun * scl + pe;
#pragma unroll
                for (int f = 0; f < NF; ++f) zacc[f] = zacc[f] * scl + pe * M[(n * P + p) * D + lane + 64 * f];
                mrun = mn;
            }
            const float inv = 1.f / lrun;
#pragma unroll
            for (int f = 0; f < NF; ++f) Z[n * D + lane + 64 * f] = zacc[f] * inv;
            if (lane < P) beta[n * P + lane] = __expf(sreg - mrun) * inv;
        }
    }
}

// slab row per block: [D*A] dW | [A] db | [A] du   (A = the FULL attention width; this launch owns
// the columns [a_off, a_off + 64))
template <int DT>
__global__ __launch_bounds__(256) void sem_attn_bwd_gen_kernel(const float *__restrict__ M, const float *Wg,
                                                               const float *bw, const float *uw,
                                                               const float *beta, const float *dZ, float *dM,
                                                               float *slab, int64_t N, int P, int A, int a_off) {
    constexpr int D = 16 * DT;
    constexpr int NF = D / 64;
    constexpr int AS = 64;           // columns of the attention space per launch
    constexpr int TA = AS / 16;
    constexpr int WLD1 = AS + 16;
    constexpr int WLD2 = AS + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *W1 = smem;                         // [D][WLD1]
    float *W2 = W1 + D * WLD1;                // [D][WLD2]
    float *dp = W2 + D * WLD2;                // [4 waves][16][WLD2]
    float *dsb = dp + 4 * 16 * WLD2;          // [2][ROWS]
    float *btb = dsb + 2 * ROWS;              // [2][ROWS]
    for (int i = threadIdx.x; i < D * AS; i += 256) {
        const float v = Wg[(int64_t)(i / AS) * A + a_off + (i % AS)];
        W1[(i / AS) * WLD1 + (i % AS)] = v;
        W2[(i / AS) * WLD2 + (i % AS)] = v;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    float bcol[TA], ucol[TA], du[TA], db[TA];
    f32x4 dW[DT][TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        bcol[t] = bw[a_off + 16 * t + l15];
        ucol[t] = uw[a_off + 16 * t + l15];
        du[t] = 0.f;
        db[t] = 0.f;
#pragma unroll
        for (int ft = 0; ft < DT; ++ft) dW[ft][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float *mydp = dp + w * 16 * WLD2;
    __syncthreads();
    const int NB = ROWS / P;
    const int64_t nchunks = (N + NB - 1) / NB;
    int buf = 0;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x, buf ^= 1) {
        const int64_t node0 = ch * NB;
        const int nodes = (int)((N - node0) < NB ? (N - node0) : NB);
        const int rows = nodes * P;
        const int64_t row0 = node0 * P;
        float *dsr = dsb + buf * ROWS, *btr = btb + buf * ROWS;
        if (threadIdx.x < ROWS && threadIdx.x >= rows) {
            dsr[threadIdx.x] = 0.f;
            btr[threadIdx.x] = 0.f;
        }
        for (int nd = w; nd < nodes; nd += 4) {
            const int64_t n = node0 + nd;
            float dz[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) dz[f] = dZ[n * D + lane + 64 * f];
            const float breg = lane < P ? beta[n * P + lane] : 0.f;
            float dbreg = 0.f;
            for (int p = 0; p < P; ++p) {
                float part = 0.f;
#pragma unroll
                for (int f = 0; f < NF; ++f) part += dz[f] * M[(n * P + p) * D + lane + 64 * f];
                const float d = han_wave_sum(part);
                if (lane == p) dbreg = d;
            }
            const float S = han_wave_sum(breg * dbreg);
            if (lane < P) {
                dsr[nd * P + lane] = breg * (dbreg - S);
                btr[nd * P + lane] = breg;
            }
        }
        __syncthreads();
        f32x4 acc[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const int lr = 16 * w + l15;
            const float *mrow = M + (row0 + (lr < rows ? lr : rows - 1)) * D + l4;
#pragma unroll 8
            for (int ks = 0; ks < D / 4; ++ks) {
                const float a = mrow[4 * ks];
                const float *wr = W1 + (4 * ks + l4) * WLD1 + l15;
#pragma unroll
                for (int t = 0; t < TA; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wr[16 * t], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float ds = dsr[16 * w + 4 * l4 + reg];
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const float v = fast_tanh(acc[t][reg] + bcol[t]);
                const float d = ds * ucol[t] * (1.f - v * v);
                du[t] += ds * v;
                db[t] += d;
                acc[t][reg] = d;
                mydp[(4 * l4 + reg) * WLD2 + 16 * t + l15] = d;
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int lr = 16 * w + 4 * l4 + reg;
            const float *mrow = M + (row0 + (lr < rows ? lr : rows - 1)) * D + l15;
#pragma unroll
            for (int ft = 0; ft < DT; ++ft) {
                const float a = mrow[16 * ft];
#pragma unroll
                for (int t = 0; t < TA; ++t)
                    dW[ft][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc[t][reg], dW[ft][t], 0, 0, 0);
            }
        }
        f32x4 acc2[DT];
#pragma unroll
        for (int ft = 0; ft < DT; ++ft) acc2[ft] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ks = 0; ks < AS / 4; ++ks) {
            const float a = mydp[l15 * WLD2 + 4 * ks + l4];
#pragma unroll
            for (int ft = 0; ft < DT; ++ft) {
                const float b = W2[(16 * ft + l15) * WLD2 + 4 * ks + l4];
                acc2[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2[ft], 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int lr = 16 * w + 4 * l4 + reg;
            if (lr < rows) {
                const int64_t n = node0 + lr / P;
                const float bt = btr[lr];
#pragma unroll
                for (int ft = 0; ft < DT; ++ft) {
                    const int f = 16 * ft + l15;
                    float *dst = dM + (row0 + lr) * D + f;
                    if (a_off == 0) *dst = acc2[ft][reg] + bt * dZ[n * D + f];
                    else *dst += acc2[ft][reg];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TA; ++t) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            du[t] += __shfl_xor(du[t], o, 64);
            db[t] += __shfl_xor(db[t], o, 64);
        }
    }
    __syncthreads();
    float *red = smem;   // [D*AS] dW | [AS] db | [AS] du  (fits inside W1 + W2)
    for (int ww = 0; ww < 4; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int ft = 0; ft < DT; ++ft)
#pragma unroll
                for (int t = 0; t < TA; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int idx = (16 * ft + 4 * l4 + reg) * AS + 16 * t + l15;
                        red[idx] = (ww == 0 ? 0.f : red[idx]) + dW[ft][t][reg];
                    }
            if (l4 == 0) {
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    const int idx = D * AS + 16 * t + l15;
                    red[idx] = (ww == 0 ? 0.f : red[idx]) + db[t];
                    red[idx + AS] = (ww == 0 ? 0.f : red[idx + AS]) + du[t];
                }
            }
        }
        __syncthreads();
    }
    float *out = slab + (int64_t)blockIdx.x * ((int64_t)D * A + 2 * A);
    for (int i = threadIdx.x; i < D * AS; i += 256) out[(int64_t)(i / AS) * A + a_off + (i % AS)] = red[i];
    for (int i = threadIdx.x; i < AS; i += 256) {
        out[(int64_t)D * A + a_off + i] = red[D * AS + i];
        out[(int64_t)D * A + A + a_off + i] = red[D * AS + AS + i];
    }
}

// ---------------------------------------------------------------------------------------------
// Any embedding width and any attention size (round 3): D and A are run-time multiples of 64 -- a last
// layer of 8 heads x 32 (D = 256), hid_units = [128] with 8 heads (D = 1024), mp_att_size = 512 ...
// (models/gat.py:42-61 leaves all of them free).  Womega (D x A, at most a few hundred KB) no longer fits
// the LDS next to the tiles, so the B fragments of the contraction come straight from global memory: every
// wave of the chip reads the same matrix, which lives in the L1s / L2s.  The forward runs the attention
// space in slices of 64*CA columns -- the score s = sum_a u_a tanh(pre_a) is additive over slices -- and
// the backward in slices of 64 columns x chunks of 128 embedding columns for dWomega (one launch per
// (slice, chunk); the launches of a slice after the first recompute pre and only add their dW chunk).
// Off the tuned path: ~L1-bandwidth-bound (5 fragment loads per 4 MFMAs), measured in DESIGN.md sec. 3.
// ---------------------------------------------------------------------------------------------
template <int CA>
__global__ __launch_bounds__(256) void sem_attn_fwd_wide_kernel(const float *__restrict__ M, const float *__restrict__ Wg,
                                                                const float *bw, const float *uw, float *Z,
                                                                float *beta, int64_t N, int P, int D, int A) {
    constexpr int AS = 64 * CA;
    constexpr int TA = AS / 16;
    __shared__ float sc[2 * ROWS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int NB = ROWS / P;
    const int64_t nchunks = (N + NB - 1) / NB;
    int buf = 0;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x, buf ^= 1) {
        const int64_t node0 = ch * NB;
        const int nodes = (int)((N - node0) < NB ? (N - node0) : NB);
        const int rows = nodes * P;
        const int64_t row0 = node0 * P;
        {
            const int lr = 16 * w + l15;
            const float *mrow = M + (row0 + (lr < rows ? lr : rows - 1)) * D + l4;
            float spart[4] = {0.f, 0.f, 0.f, 0.f};
            for (int a_off = 0; a_off < A; a_off += AS) {
                f32x4 acc[TA];
#pragma unroll
                for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float *wcol = Wg + (int64_t)l4 * A + a_off + l15;
#pragma unroll 4
                for (int ks = 0; ks < D / 4; ++ks) {
                    const float a = mrow[4 * ks];
                    const float *wr = wcol + (int64_t)(4 * ks) * A;
#pragma unroll
                    for (int t = 0; t < TA; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wr[16 * t], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    const float bc = bw[a_off + 16 * t + l15], uc = uw[a_off + 16 * t + l15];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) spart[reg] += fast_tanh(acc[t][reg] + bc) * uc;
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float s = han_row16_sum(spart[reg]);
                if (l15 == 0) sc[buf * ROWS + 16 * w + 4 * l4 + reg] = s;
            }
        }
        __syncthreads();
        for (int nd = w; nd < nodes; nd += 4) {
            const int64_t n = node0 + nd;
            const float *srow = sc + buf * ROWS + nd * P;
            float mrun = HAN_NEG_BIG;
            for (int p = 0; p < P; ++p) mrun = fmaxf(mrun, srow[p]);
            float lrun = 0.f;
            for (int p = 0; p < P; ++p) lrun += __expf(srow[p] - mrun);
            const float inv = 1.f / lrun;
            for (int f = lane; f < D; f += 64) {
                float z = 0.f;
                for (int p = 0; p < P; ++p) z += __expf(srow[p] - mrun) * M[(n * P + p) * D + f];
                Z[n * D + f] = z * inv;
            }
            if (lane < P) beta[n * P + lane] = __expf(srow[lane] - mrun) * inv;
        }
    }
}

constexpr int kWideDChunk = 128;     // embedding columns of dWomega per backward launch (DTC = 8 tiles)

// slab row per block: [D*A] dW | [A] db | [A] du; this launch owns the attention columns [a_off, a_off + 64) and the
// dW rows [d_off, d_off + 128); the launch with d_off == 0 also owns db / du of its slice and adds the slice's dM term
__global__ __launch_bounds__(256) void sem_attn_bwd_wide_kernel(const float *__restrict__ M, const float *__restrict__ Wg,
                                                                const float *bw, const float *uw,
                                                                const float *beta, const float *dZ, float *dM,
                                                                float *slab, int64_t N, int P, int D, int A,
                                                                int a_off, int d_off) {
    constexpr int AS = 64;
    constexpr int TA = AS / 16;
    constexpr int DTC = kWideDChunk / 16;
    constexpr int WLD2 = AS + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *dp = smem;                         // [4 waves][16][WLD2]
    float *dsb = dp + 4 * 16 * WLD2;          // [2][ROWS]
    float *btb = dsb + 2 * ROWS;              // [2][ROWS]
    const bool first = d_off == 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ntile = (D - d_off) / 16 < DTC ? (D - d_off) / 16 : DTC;   // dW tiles of this chunk that exist
    float bcol[TA], ucol[TA], du[TA], db[TA];
    f32x4 dW[DTC][TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        bcol[t] = bw[a_off + 16 * t + l15];
        ucol[t] = uw[a_off + 16 * t + l15];
        du[t] = 0.f;
        db[t] = 0.f;
#pragma unroll
        for (int ft = 0; ft < DTC; ++ft) dW[ft][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float *mydp = dp + w * 16 * WLD2;
    const int NB = ROWS / P;
    const int64_t nchunks = (N + NB - 1) / NB;
    int buf = 0;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x, buf ^= 1) {
        const int64_t node0 = ch * NB;
        const int nodes = (int)((N - node0) < NB ? (N - node0) : NB);
        const int rows = nodes * P;
        const int64_t row0 = node0 * P;
        float *dsr = dsb + buf * ROWS, *btr = btb + buf * ROWS;
        if (threadIdx.x < ROWS && threadIdx.x >= rows) {
            dsr[threadIdx.x] = 0.f;
            btr[threadIdx.x] = 0.f;
        }
        for (int nd = w; nd < nodes; nd += 4) {
            const int64_t n = node0 + nd;
            const float breg = lane < P ? beta[n * P + lane] : 0.f;
            float dbreg = 0.f;
            for (int p = 0; p < P; ++p) {
                float part = 0.f;
                for (int f = lane; f < D; f += 64) part += dZ[n * D + f] * M[(n * P + p) * D + f];
                const float d = han_wave_sum(part);
                if (lane == p) dbreg = d;
            }
            const float S = han_wave_sum(breg * dbreg);
            if (lane < P) {
                dsr[nd * P + lane] = breg * (dbreg - S);
                btr[nd * P + lane] = breg;
            }
        }
        __syncthreads();
        f32x4 acc[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const int lr = 16 * w + l15;
            const float *mrow = M + (row0 + (lr < rows ? lr : rows - 1)) * D + l4;
            const float *wcol = Wg + (int64_t)l4 * A + a_off + l15;
#pragma unroll 4
            for (int ks = 0; ks < D / 4; ++ks) {
                const float a = mrow[4 * ks];
                const float *wr = wcol + (int64_t)(4 * ks) * A;
#pragma unroll
                for (int t = 0; t < TA; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wr[16 * t], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float ds = dsr[16 * w + 4 * l4 + reg];
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const float v = fast_tanh(acc[t][reg] + bcol[t]);
                const float d = ds * ucol[t] * (1.f - v * v);
                du[t] += ds * v;
                db[t] += d;
                acc[t][reg] = d;
                if (first) mydp[(4 * l4 + reg) * WLD2 + 16 * t + l15] = d;
            }
        }
        // G3: dW[d_off + 16 ft + .][a_off + 16 t + .] += M^T . dpre over this tile's rows
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int lr = 16 * w + 4 * l4 + reg;
            const float *mrow = M + (row0 + (lr < rows ? lr : rows - 1)) * D + d_off + l15;
#pragma unroll
            for (int ft = 0; ft < DTC; ++ft) {
                if (ft < ntile) {
                    const float a = mrow[16 * ft];
#pragma unroll
                    for (int t = 0; t < TA; ++t)
                        dW[ft][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc[t][reg], dW[ft][t], 0, 0, 0);
                }
            }
        }
        // G2 (first launch of the slice): dM[r][d] (+)= sum_a dpre[r][a] Womega[d][a], four 16-column tiles at a time
        if (first) {
            for (int dt0 = 0; dt0 < D / 16; dt0 += 4) {
                f32x4 acc2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float *wrow = Wg + (int64_t)(16 * dt0 + l15) * A + a_off + l4;
#pragma unroll 4
                for (int ks = 0; ks < AS / 4; ++ks) {
                    const float a = mydp[l15 * WLD2 + 4 * ks + l4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wrow[(int64_t)(16 * j) * A + 4 * ks], acc2[j], 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int lr = 16 * w + 4 * l4 + reg;
                    if (lr < rows) {
                        const int64_t n = node0 + lr / P;
                        const float bt = btr[lr];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int f = 16 * (dt0 + j) + l15;
                            float *dst = dM + (row0 + lr) * D + f;
                            if (a_off == 0) *dst = acc2[j][reg] + bt * dZ[n * D + f];
                            else *dst += acc2[j][reg];
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TA; ++t) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            du[t] += __shfl_xor(du[t], o, 64);
            db[t] += __shfl_xor(db[t], o, 64);
        }
    }
    __syncthreads();
    float *red = smem;   // [128*AS] dW | [AS] db | [AS] du
    for (int ww = 0; ww < 4; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int ft = 0; ft < DTC; ++ft)
#pragma unroll
                for (int t = 0; t < TA; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int idx = (16 * ft + 4 * l4 + reg) * AS + 16 * t + l15;
                        red[idx] = (ww == 0 ? 0.f : red[idx]) + dW[ft][t][reg];
                    }
            if (l4 == 0) {
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    const int idx = kWideDChunk * AS + 16 * t + l15;
                    red[idx] = (ww == 0 ? 0.f : red[idx]) + db[t];
                    red[idx + AS] = (ww == 0 ? 0.f : red[idx + AS]) + du[t];
                }
            }
        }
        __syncthreads();
    }
    float *out = slab + (int64_t)blockIdx.x * ((int64_t)D * A + 2 * A);
    for (int i = threadIdx.x; i < 16 * ntile * AS; i += 256)
        out[(int64_t)(d_off + i / AS) * A + a_off + (i % AS)] = red[i];
    if (first)
        for (int i = threadIdx.x; i < AS; i += 256) {
            out[(int64_t)D * A + a_off + i] = red[kWideDChunk * AS + i];
            out[(int64_t)D * A + A + a_off + i] = red[kWideDChunk * AS + AS + i];
        }
}

constexpr int kSemBwdBlocks = 256;   // one 4-wave block per CU (104 KB of LDS at A = 128)

template <int CA>
size_t fwd_lds() { return (size_t)(64 * (64 * CA + 16) + 2 * ROWS) * sizeof(float); }
template <int CA>
size_t bwd_lds() {
    constexpr int A = 64 * CA;
    return (size_t)(64 * (A + 16) + 64 * (A + 2) + 4 * 16 * (A + 2) + 4 * ROWS) * sizeof(float);
}

template <int CA>
int launch_fwd(const float *M, const float *w, const float *b, const float *u, float *Z, float *beta,
               int64_t N, int P, int flags, hipStream_t st) {
    const size_t lds = fwd_lds<CA>();
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)sem_attn_fwd_kernel<CA>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    if ((P == 1 || P == 2 || P == 4 || P == 8 || P == 16) && N * P >= 64 * 1024 && !(flags & HAN_FLAG_K3_EXACT_PIPE)) {
        // large inputs: the contraction on the bf16 matrix pipe (exact 3-way split, fp32-class accuracy)
        const size_t blds = (size_t)3 * 64 * CA * 128;       // swizzled 128-B rows (sem_attn_fwd_wave_b6_kernel)
        const int grid = han_grid_for(N * P, 64, 256 * 3);
        hipError_t e2 = hipSuccess;
#define HAN_LAUNCH_FWD_B6(PV)                                                                            \
    e2 = hipFuncSetAttribute((const void *)sem_attn_fwd_wave_b6_kernel<CA, PV>,                          \
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds);                     \
    if (e2 == hipSuccess) sem_attn_fwd_wave_b6_kernel<CA, PV><<<grid, 256, blds, st>>>(M, w, b, u, Z, beta, N);
        switch (P) {
            case 1: HAN_LAUNCH_FWD_B6(1) break;
            case 2: HAN_LAUNCH_FWD_B6(2) break;
            case 4: HAN_LAUNCH_FWD_B6(4) break;
            case 8: HAN_LAUNCH_FWD_B6(8) break;
            default: HAN_LAUNCH_FWD_B6(16) break;
        }
#undef HAN_LAUNCH_FWD_B6
        if (e2 != hipSuccess) return (int)e2;
        HAN_CHECK_LAUNCH();
        return 0;
    }
    if (P == 1 || P == 2 || P == 4 || P == 8 || P == 16) {
        constexpr bool WREG = false;
        const size_t wlds = WREG ? 0 : (size_t)64 * (64 * CA + 16) * sizeof(float);
        const int grid = han_grid_for(N * P, 64, 256 * (WREG ? 2 : 4));
        if (P == 1) sem_attn_fwd_wave_kernel<CA, 1, WREG><<<grid, 256, wlds, st>>>(M, w, b, u, Z, beta, N);
        else if (P == 2) sem_attn_fwd_wave_kernel<CA, 2, WREG><<<grid, 256, wlds, st>>>(M, w, b, u, Z, beta, N);
        else if (P == 4) sem_attn_fwd_wave_kernel<CA, 4, WREG><<<grid, 256, wlds, st>>>(M, w, b, u, Z, beta, N);
        else if (P == 8) sem_attn_fwd_wave_kernel<CA, 8, WREG><<<grid, 256, wlds, st>>>(M, w, b, u, Z, beta, N);
        else sem_attn_fwd_wave_kernel<CA, 16, WREG><<<grid, 256, wlds, st>>>(M, w, b, u, Z, beta, N);
        HAN_CHECK_LAUNCH();
        return 0;
    }
    const int NB = ROWS / P;
    const int grid = han_grid_for(N, NB, 256 * 3);
    sem_attn_fwd_kernel<CA><<<grid, 256, lds, st>>>(M, w, b, u, Z, beta, N, P);
    HAN_CHECK_LAUNCH();
    return 0;
}

template <int CA>
int launch_bwd(const float *M, const float *w, const float *b, const float *u, const float *beta,
               const float *dZ, float *dM, float *slab, int64_t N, int P, int *grid_out, int flags, hipStream_t st) {
    const size_t lds = bwd_lds<CA>();
    hipError_t e = hipFuncSetAttribute((const void *)sem_attn_bwd_kernel<CA>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    if ((P == 1 || P == 2 || P == 4 || P == 8 || P == 16) && N * P >= 64 * 1024 && !(flags & HAN_FLAG_K3_EXACT_PIPE)) {
        const int grid = han_grid_for(N * P, 64, kSemBwdBlocks);
        *grid_out = grid;
        constexpr int A6 = 64 * CA;
        const size_t blds = (size_t)3 * A6 * SA_WLDB + (size_t)3 * 64 * (A6 * 2 + 32) + (size_t)(4 * 16 * (A6 + 4) + 2 * A6) * sizeof(float);
        hipError_t e3 = hipSuccess;
#define HAN_LAUNCH_BWD_B6_AS(KERNEL, THREADS)                                                            \
    e3 = hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds); \
    if (e3 == hipSuccess) KERNEL<<<grid, THREADS, blds, st>>>(M, w, b, u, beta, dZ, dM, slab, N);
#define HAN_LAUNCH_BWD_B6(PV)                                                                            \
    if (CA == 2 && (flags & HAN_FLAG_K3_PAIRS)) {                                                        \
        HAN_LAUNCH_BWD_B6_AS((sem_attn_bwd_pair_b6_kernel<PV>), 512)                                     \
    } else if (flags & HAN_FLAG_K3_G3_F32) {                                                             \
        HAN_LAUNCH_BWD_B6_AS((sem_attn_bwd_wave_b6_kernel<CA, PV, false>), 256)                          \
    } else {                                                                                             \
        HAN_LAUNCH_BWD_B6_AS((sem_attn_bwd_wave_b6_kernel<CA, PV, true>), 256)                           \
    }
        switch (P) {
            case 1: HAN_LAUNCH_BWD_B6(1) break;
            case 2: HAN_LAUNCH_BWD_B6(2) break;
            case 4: HAN_LAUNCH_BWD_B6(4) break;
            case 8: HAN_LAUNCH_BWD_B6(8) break;
            default: HAN_LAUNCH_BWD_B6(16) break;
        }
#undef HAN_LAUNCH_BWD_B6
#undef HAN_LAUNCH_BWD_B6_AS
        if (e3 != hipSuccess) return (int)e3;
        HAN_CHECK_LAUNCH();
        return 0;
    }
    if (P == 1 || P == 2 || P == 4 || P == 8 || P == 16) {
        const int grid = han_grid_for(N > 0 ? N * P : 1, 64, kSemBwdBlocks);
        *grid_out = grid;
        hipError_t e2 = hipSuccess;
#define HAN_LAUNCH_BWD_WAVE(PV)                                                                          \
    e2 = hipFuncSetAttribute((const void *)sem_attn_bwd_wave_kernel<CA, PV>,                             \
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
    if (e2 == hipSuccess)                                                                                \
        sem_attn_bwd_wave_kernel<CA, PV><<<grid, 256, lds, st>>>(M, w, b, u, beta, dZ, dM, slab, N);
        switch (P) {
            case 1: HAN_LAUNCH_BWD_WAVE(1) break;
            case 2: HAN_LAUNCH_BWD_WAVE(2) break;
            case 4: HAN_LAUNCH_BWD_WAVE(4) break;
            case 8: HAN_LAUNCH_BWD_WAVE(8) break;
            default: HAN_LAUNCH_BWD_WAVE(16) break;
        }
#undef HAN_LAUNCH_BWD_WAVE
        if (e2 != hipSuccess) return (int)e2;
        HAN_CHECK_LAUNCH();
        return 0;
    }
    const int NB = ROWS / P;
    const int grid = han_grid_for(N > 0 ? N : 1, NB, kSemBwdBlocks);
    *grid_out = grid;
    sem_attn_bwd_kernel<CA><<<grid, 256, lds, st>>>(M, w, b, u, beta, dZ, dM, slab, N, P);
    HAN_CHECK_LAUNCH();
    return 0;
}

template <int CA, int DT>
int launch_fwd_gen(const float *M, const float *w, const float *b, const float *u, float *Z, float *beta,
                   int64_t N, int P, hipStream_t st) {
    const size_t lds = (size_t)(16 * DT * (64 * CA + 16) + 2 * ROWS) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void *)sem_attn_fwd_gen_kernel<CA, DT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int grid = han_grid_for(N, ROWS / P, 256 * 2);
    sem_attn_fwd_gen_kernel<CA, DT><<<grid, 256, lds, st>>>(M, w, b, u, Z, beta, N, P);
    HAN_CHECK_LAUNCH();
    return 0;
}

template <int DT>
int launch_bwd_gen(const float *M, const float *w, const float *b, const float *u, const float *beta,
                   const float *dZ, float *dM, float *slab, int64_t N, int P, int A, int *grid_out,
                   hipStream_t st) {
    constexpr int D = 16 * DT;
    const size_t lds = (size_t)(D * (64 + 16) + D * (64 + 2) + 4 * 16 * (64 + 2) + 4 * ROWS) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void *)sem_attn_bwd_gen_kernel<DT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int grid = han_grid_for(N > 0 ? N : 1, ROWS / P, kSemBwdBlocks);
    *grid_out = grid;
    for (int a_off = 0; a_off < A; a_off += 64) {      // same grid every pass: a block's slab row fills up slice by slice
        sem_attn_bwd_gen_kernel<DT><<<grid, 256, lds, st>>>(M, w, b, u, beta, dZ, dM, slab, N, P, A, a_off);
        HAN_CHECK_LAUNCH();
    }
    return 0;
}

template <int CA>
int launch_fwd_wide(const float *M, const float *w, const float *b, const float *u, float *Z, float *beta,
                    int64_t N, int P, int D, int A, hipStream_t st) {
    const int grid = han_grid_for(N, ROWS / P, 256 * 4);
    sem_attn_fwd_wide_kernel<CA><<<grid, 256, 0, st>>>(M, w, b, u, Z, beta, N, P, D, A);
    HAN_CHECK_LAUNCH();
    return 0;
}

int launch_bwd_wide(const float *M, const float *w, const float *b, const float *u, const float *beta,
                    const float *dZ, float *dM, float *slab, int64_t N, int P, int D, int A, int *grid_out,
                    hipStream_t st) {
    const size_t lds = (size_t)(kWideDChunk * 64 + 2 * 64) * sizeof(float);
    const int grid = han_grid_for(N > 0 ? N : 1, ROWS / P, kSemBwdBlocks);
    *grid_out = grid;
    for (int a_off = 0; a_off < A; a_off += 64)          // same grid every pass: a block's slab row fills up piece by piece
        for (int d_off = 0; d_off < D; d_off += kWideDChunk) {
            sem_attn_bwd_wide_kernel<<<grid, 256, lds, st>>>(M, w, b, u, beta, dZ, dM, slab, N, P, D, A, a_off, d_off);
            HAN_CHECK_LAUNCH();
        }
    return 0;
}

// shapes of the tuned kernels; everything else (multiples of 64) goes to the run-time-width kernels
bool k3_tuned(int D, int A) { return (D == 64 || D == 128) && A >= 64 && A <= 256; }
bool k3_shape_ok(int P, int D, int A) { return P >= 1 && P <= 64 && D >= 64 && D % 64 == 0 && A >= 64 && A % 64 == 0; }

}  // namespace

extern "C" int han_sem_attn_fwd(const float *M, const float *w_omega, const float *b_omega,
                                const float *u_omega, float *Z, float *beta, int64_t N, int P, int D, int A,
                                int flags, void *stream) {
    if (N == 0) return 0;   // nothing to do; row pointers of empty tensors may be null
    if (!M || !w_omega || !b_omega || !u_omega || !Z || !beta || N < 0 || P <= 0) return HAN_E_BADARG;
    if (!k3_shape_ok(P, D, A)) return HAN_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (!k3_tuned(D, A)) {      // any wider embedding / attention space
        if (A % 256 == 0) return launch_fwd_wide<4>(M, w_omega, b_omega, u_omega, Z, beta, N, P, D, A, st);
        if (A % 192 == 0) return launch_fwd_wide<3>(M, w_omega, b_omega, u_omega, Z, beta, N, P, D, A, st);
        if (A % 128 == 0) return launch_fwd_wide<2>(M, w_omega, b_omega, u_omega, Z, beta, N, P, D, A, st);
        return launch_fwd_wide<1>(M, w_omega, b_omega, u_omega, Z, beta, N, P, D, A, st);
    }
    // attention spaces of 192 / 256 columns (round 3) and 128-wide embeddings: the width-templated block-level kernels
    if (D == 128 || A > 128) {
        if (D == 128) {
            switch (A / 64) {
                case 1: return launch_fwd_gen<1, 8>(M, w_omega, b_omega, u_omega, Z, beta, N, P, st);
                case 2: return launch_fwd_gen<2, 8>(M, w_omega, b_omega, u_omega, Z, beta, N, P, st);
                case 3: return launch_fwd_gen<3, 8>(M, w_omega, b_omega, u_omega, Z, beta, N, P, st);
                default: return launch_fwd_gen<4, 8>(M, w_omega, b_omega, u_omega, Z, beta, N, P, st);
            }
        }
        if (A == 192) return launch_fwd_gen<3, 4>(M, w_omega, b_omega, u_omega, Z, beta, N, P, st);
        return launch_fwd_gen<4, 4>(M, w_omega, b_omega, u_omega, Z, beta, N, P, st);
    }
    if (A == 64) return launch_fwd<1>(M, w_omega, b_omega, u_omega, Z, beta, N, P, flags, st);
    return launch_fwd<2>(M, w_omega, b_omega, u_omega, Z, beta, N, P, flags, st);
}

extern "C" size_t han_sem_attn_bwd_workspace(int64_t N, int P, int D, int A) {
    (void)N; (void)P;
    return (size_t)kSemBwdBlocks * (size_t)(D * A + 2 * A) * sizeof(float);
}

extern "C" int han_sem_attn_bwd(const float *M, const float *w_omega, const float *b_omega,
                                const float *u_omega, const float *beta, const float *dZ, float *dM,
                                float *dw_omega, float *db_omega, float *du_omega, void *workspace,
                                size_t workspace_bytes, int64_t N, int P, int D, int A, int flags, void *stream) {
    if (!M || !w_omega || !b_omega || !u_omega || !beta || !dZ || !dM || !dw_omega || !db_omega || !du_omega ||
        !workspace || N < 0 || P <= 0)
        return HAN_E_BADARG;
    if (!k3_shape_ok(P, D, A)) return HAN_E_UNSUPPORTED;
    if (workspace_bytes < han_sem_attn_bwd_workspace(N, P, D, A)) return HAN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *slab = (float *)workspace;
    int grid = 0;
    int rc;
    if (!k3_tuned(D, A)) rc = launch_bwd_wide(M, w_omega, b_omega, u_omega, beta, dZ, dM, slab, N, P, D, A, &grid, st);
    else if (D == 128) rc = launch_bwd_gen<8>(M, w_omega, b_omega, u_omega, beta, dZ, dM, slab, N, P, A, &grid, st);
    else if (A > 128) rc = launch_bwd_gen<4>(M, w_omega, b_omega, u_omega, beta, dZ, dM, slab, N, P, A, &grid, st);
    else rc = (A == 64)
                 ? launch_bwd<1>(M, w_omega, b_omega, u_omega, beta, dZ, dM, slab, N, P, &grid, flags, st)
                 : launch_bwd<2>(M, w_omega, b_omega, u_omega, beta, dZ, dM, slab, N, P, &grid, flags, st);
    if (rc != 0) return rc;
    const int width = D * A + 2 * A;
    HanReduceOut o = han_reduce_to(dw_omega, width);
    o.ptr[1] = db_omega; o.ptr[2] = du_omega;
    o.seg_end[0] = D * A; o.seg_end[1] = D * A + A; o.seg_end[2] = width;
    o.nseg = 3;
    hipError_t e = han_reduce_slabs(slab, grid, width, width, o, st);
    if (e != hipSuccess) return (int)e;
    return 0;
}
