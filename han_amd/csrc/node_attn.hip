// K2 -- node-level attention over CSR neighbours, forward and backward (gfx950).
//
// Reference arithmetic: utils/layers.py:26-35,46 (attn_head, dense additive
// mask) == utils/layers.py:95-118,127 (sp_attn_head) restricted to stored
// neighbours.  The N x N logits / coefficient tensors of the reference are never
// materialised.
//
// Data layout (HBM): a projected row H_j is D = 64 fp32 = 256 B = two 128-B
// lines.  Lane mapping: a wave is 4 groups of 16 lanes; lane q of a group owns
// features 4q..4q+3 (one dwordx4 = 16 B per lane, 256 B per group per load
// instruction), i.e. head (4q)/FP.  The 4 groups take 4 different neighbours of
// the same destination row per step and U steps are kept in flight, so one wave
// has 4*U gathered rows (4 KiB at U = 4) outstanding.
//
// The neighbour's score f2_j = H_j[k,:] . a2[k] + b2[k] (layers.py:24) is
// recomputed from the gathered row (4 FMAs + one DPP add per edge) instead of
// being gathered from a second table: a 32-B gather costs a whole extra memory
// sector per edge (measured round 1: 19.4 GB of fabric traffic per launch against
// 14.9 GB algorithmic).  In training the gathered rows are the UNDROPPED H; the
// Bernoulli draw of the projected-row dropout (layers.py:31-32) travels IN the row:
// K1 stores each element's keep bit in the lowest mantissa bit of the float (a
// 1-ulp perturbation that every consumer sees consistently), so the mask costs no
// extra gather; it is applied after the score was taken, as the reference orders it.
//
// Softmax is an online (running max / running sum) softmax per lane; the 4 groups'
// partial (m, l, acc) are merged with two xor-shuffles (16, 32) at the end of the row.
//
// Roofline: HBM / Infinity-Cache gather bandwidth.  Algorithmic bytes per edge
// (SURVEY.md sec. 8d): 4 (colidx) + 256 (H_j) + 4K (f2_j) = 292 B at K = 8.
#include "han_common.h"

namespace {

// sum over the FP/4 lanes that share a head (contiguous, aligned lanes);
// DPP quad permutes for the first two steps (pure VALU), bpermute beyond
template <int FP>
__device__ __forceinline__ float head_sum(float v) {
    if (FP >= 8)
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));  // xor 1
    if (FP >= 16)
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));  // xor 2
    if (FP >= 32) v += __shfl_xor(v, 4, 64);
    if (FP >= 64) v += __shfl_xor(v, 8, 64);
    return v;
}

__device__ __forceinline__ float dot4(const float4_t &x, const float4_t &y) {
    return x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
}

// XCD-aware work order.  Blocks are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each XCD
// has its own 4 MB L2), and a grid-stride sweep keeps ~gridDim.x * 4 consecutive rows in flight.  With the
// identity order every XCD touches rows spread over that whole stripe, so on a graph with locality all eight
// L2s cache the SAME source window (each source row is fetched up to 8 times per sweep).  Here the blocks of
// one XCD take one contiguous eighth of every stripe instead: the eight L2s hold eight different windows.
// Pure speed: any bijection of the work units is correct; it is the identity when gridDim.x % 8 != 0 and when
// the caller does not set HAN_FLAG_XCD_ORDER (graphs without locality: 9 % slower in the HBM regime).
__device__ __forceinline__ int64_t xcd_first_unit(int wave_in_block, int xcd_order) {
    const unsigned b = blockIdx.x, g = gridDim.x;
    const unsigned vb = (xcd_order && g % 8u == 0u) ? (b % 8u) * (g / 8u) + b / 8u : b;
    return (int64_t)vb * 4 + wave_in_block;
}

struct FwdArgs {
    const int64_t *rowptr;
    const int32_t *colidx;
    const float *edge_val;   // stored adjacency values scaling the logits (layers.py:95-96), or null (binary)
    const void *H;  // fp32 (256-B rows) or bf16 (128-B rows) table
    const int32_t *gid;   // global id of each table row (halo tables), or null: the index is the id
    int lsb_mask;   // training with fts dropout: the lowest mantissa bit of every H element is its keep bit
    const float *f1;
    const float *f2g;   // F' = 64 only: gathered neighbour scores (slices of a wide head), or null: recomputed from the row
    const float *a2;
    const float *b2;
    const float *c;
    const float *res;   // residual term added before the activation (layers.py:38-40), or null
    float *out;
    int64_t out_stride;
    float *pre, *lse, *aggp, *tsum;
    int64_t N;
    // degree-binned launches (han_row_split_t: short_rows / mid_rows): this launch covers the n_work rows listed in
    // `rows` (null: the rows 0 .. n_work-1 themselves)
    const int32_t *rows;
    int64_t n_work;
    const int *only_if;      // lean kernels behind the dense path: run only when this device word is non-zero (null: always)
    int deep;                // HAN_FLAG_K2_DEEP (measurements)
    int shared_hash;         // HAN_FLAG_K2_SHARED_HASH: the shared attention-dropout hash at any size
    float slope;
    uint32_t seed_lo, seed_hi, thr_coef;
    const uint64_t *seed_dev;
    float inv_keep_coef, inv_keep_fts;
    int64_t row_offset;
    int activation;
    int xcd_order;      // HAN_FLAG_XCD_ORDER
    // row splitting for skewed graphs (see HanRowSplit): rows longer than split_deg are
    // skipped by the main launch and handled by the partial + finish launches
    int64_t split_deg;
    int64_t n_long, n_chunks;
    const int64_t *long_rows, *long_ptr, *chunk_start, *chunk_end;
    const int32_t *chunk_long;
    float *split_ws;
};

// Backward tables: ONE fused row per destination i, [ g_i : D elements (fp32 or bf16) | (f1, lse, s, 0) x K : fp32 ],
// padded to whole 128-B lines (fp32, K = 8: 256 + 128 = 384 B = 3 lines; bf16: 128 + 128 = 256 B), so that a
// node partition moves ONE table per meta-path in the backward (one collective, one pack) instead of two.
template <int FP, bool BF>
struct GsRow {
    static constexpr int K = HAN_D / FP;
    static constexpr int g_bytes = HAN_D * (BF ? 2 : 4);
    static constexpr int bytes = ((g_bytes + 16 * K + 127) / 128) * 128;
};

template <int FP, bool BF>
__device__ __forceinline__ float4_t gs_load_g4(const void *gs, int64_t row, int q) {
    const char *base = reinterpret_cast<const char *>(gs) + row * GsRow<FP, BF>::bytes;
    if (BF) {
        const uint2 w = *reinterpret_cast<const uint2 *>(base + 8 * q);
        float4_t v;
        v[0] = __uint_as_float(w.x << 16);
        v[1] = __uint_as_float(w.x & 0xFFFF0000u);
        v[2] = __uint_as_float(w.y << 16);
        v[3] = __uint_as_float(w.y & 0xFFFF0000u);
        return v;
    }
    return *reinterpret_cast<const float4_t *>(base + 16 * q);
}

template <int FP, bool BF>
__device__ __forceinline__ float4_t gs_load_stats(const void *gs, int64_t row, int head) {
    return *reinterpret_cast<const float4_t *>(reinterpret_cast<const char *>(gs) + row * GsRow<FP, BF>::bytes +
                                               GsRow<FP, BF>::g_bytes + 16 * head);
}

constexpr int kFwdChunkStride = 192;   // floats per chunk: acc[64] | accp[64] | m[K] | l[K] | tl[K]
constexpr int kBwdChunkStride = 80;    // acc[64] | df2[K]

template <bool TRAIN>
struct RowState {
    float m, l, tl;
    float acc[4], accp[4];
    __device__ __forceinline__ void init() {
        m = HAN_NEG_BIG; l = 0.f; tl = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] = 0.f; accp[t] = 0.f; }
    }
    // merge the state of lane `lane ^ off`
    __device__ __forceinline__ void merge(int off) {
        const float m_o = __shfl_xor(m, off, 64);
        const float l_o = __shfl_xor(l, off, 64);
        const float M = fmaxf(m, m_o);
        const float sa = __expf(m - M), sb = __expf(m_o - M);
        l = l * sa + l_o * sb;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = acc[t] * sa + __shfl_xor(acc[t], off, 64) * sb;
        if (TRAIN) {
            tl = tl * sa + __shfl_xor(tl, off, 64) * sb;
#pragma unroll
            for (int t = 0; t < 4; ++t) accp[t] = accp[t] * sa + __shfl_xor(accp[t], off, 64) * sb;
        }
        m = M;
    }
};

// sign-extended bit `BIT` of x (0 or -1) as ONE v_bfe_i32: the compiler would otherwise
// re-canonicalise `x & sext(bit)` into and + compare + select
template <int BIT>
__device__ __forceinline__ int han_bit_mask(int x) {
    int r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(x), "n"(BIT));
    return r;
}

// the value lane `u` of this lane's DPP quad (4 consecutive lanes) holds: quad_perm [u, u, u, u], a full-rate move
// without an LDS round trip
__device__ __forceinline__ uint32_t quad_bcast(const uint32_t v, const int u) {
    switch (u) {
        case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, true);
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xF, 0xF, true);
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xF, 0xF, true);
        default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xF, 0xF, true);
    }
}

// Gather U neighbour rows and fold them into the running softmax state.
// FAST (training launches with both dropouts on and table index == global id): no uniform
// branch is left inside the edge loop.  ALLV: all U slots hold real edges (full steps), so
// the validity selects vanish; the tail of a row runs with U = 1.
// DD (full 4-edge steps of the FAST training launch; HAN_FLAG_K2_SHARED_HASH, a measurement form): the attention-dropout
// hash of an edge serves four heads and the four lanes of a DPP quad (q = 4m .. 4m+3) sit in ONE head quad, so lane q
// hashes edge q & 3 of the step for that head quad -- one hash per lane and step instead of four -- and takes the other
// three edges' words from its quad neighbours by quad_perm broadcasts (round 4 first handed them around with
// ds_bpermute).  Same keys, same fields: bitwise the per-lane draws.  Neither form is faster than hashing per lane
// (launch_fwd_rows): the hash is not what the training forward waits for.
template <int FP, bool TRAIN, int U, bool BF, bool VAL, bool FAST, bool ALLV, bool DD = false>
__device__ __forceinline__ void consume_edges(const FwdArgs &a, const int (&j)[U], const float (&w)[U],
                                              const bool (&valid)[U], const float f1h, const uint32_t gi, const int q, const int head,
                                              const float4_t &a24, const float b2h, const bool drop_c,
                                              RowState<TRAIN> &st) {
    constexpr int KQ = (HAN_D / FP + 3) / 4;
    static_assert(!DD || (U == 4 && TRAIN && FAST && ALLV), "the shared hash is built for full 4-edge training steps");
    uint32_t dd_x = 0, dd_y = 0;
    if (DD) {
        const int uu = q & 3;
        const int ju = uu == 0 ? j[0] : (uu == 1 ? j[1] : (uu == 2 ? j[2] : j[U - 1]));
        const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF, gi,
                                        (uint32_t)ju * (uint32_t)KQ + (uint32_t)(head >> 2));
        if constexpr (FP >= 8) dd_x = (head & 2) ? rn.y : rn.x;      // the word the whole DPP quad reads (head & 2 is uniform in it)
        else { dd_x = rn.x; dd_y = rn.y; }
    }
    float4_t hv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) hv[u] = han_load_row4<BF>(a.H, (int64_t)j[u], q);
    float ev[U], sg[U];
    float mc = st.m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // layers.py:24,26; sp_attn_head: adj_ij*f1_i + adj_ij*f2_j (:95-96), w == 1 when binary
        float x;
        if (FP == HAN_D && !FAST && a.f2g) x = f1h + a.f2g[j[u]];      // one head of 64 columns: K = 1, index = row
        else x = f1h + (head_sum<FP>(dot4(hv[u], a24)) + b2h);
        if (VAL) x *= w[u];
        sg[u] = x > 0.f ? 1.f : a.slope;
        if (VAL) sg[u] *= w[u];
        ev[u] = (ALLV || valid[u]) ? han_lrelu(x, a.slope) : HAN_NEG_BIG;          // layers.py:27
        mc = fmaxf(mc, ev[u]);
    }
    const float sc = __expf(st.m - mc);
    st.l *= sc;
#pragma unroll
    for (int t = 0; t < 4; ++t) st.acc[t] *= sc;
    if (TRAIN) {
        st.tl *= sc;
#pragma unroll
        for (int t = 0; t < 4; ++t) st.accp[t] *= sc;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float p = (ALLV || valid[u]) ? __expf(ev[u] - mc) : 0.f;
        st.l += p;
        float pd = p;
        if (TRAIN) {
            // The 1/keep factors of both dropouts are applied once per row in write_row,
            // not per edge: here a dropped term is simply zeroed.
            if (DD) {               // attention dropout from the quad's shared hashes
                uint32_t wsel;
                if constexpr (FP >= 8) wsel = quad_bcast(dd_x, u);
                else {
                    const uint32_t wx = quad_bcast(dd_x, u), wy = quad_bcast(dd_y, u);
                    wsel = (head & 2) ? wy : wx;
                }
                const uint32_t fld = (head & 1) ? (wsel >> 16) : (wsel & 0xFFFFu);
                pd = fld < a.thr_coef ? p : 0.f;
            } else if (FAST || drop_c) {   // attention dropout, layers.py:29-30
                const uint32_t gj = (!FAST && a.gid) ? (uint32_t)a.gid[j[u]] : (uint32_t)j[u];
                const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF, gi,
                                                gj * (uint32_t)KQ + (uint32_t)(head >> 2));
                pd = rn.field(head & 3) < a.thr_coef ? p : 0.f;
            }
            // projected-row dropout, layers.py:31-32 (after the score was taken): AND the
            // element with the sign-extended keep bit (one v_bfe_i32 + one v_and per element)
            if (FAST || a.lsb_mask) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int bits = __float_as_int(hv[u][t]);
                    hv[u][t] = __int_as_float(bits & han_bit_mask<BF ? 16 : 0>(bits));
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) st.acc[t] += pd * hv[u][t];
        if (TRAIN) {
            st.tl += p * sg[u];
            const float pds = pd * sg[u];
#pragma unroll
            for (int t = 0; t < 4; ++t) st.accp[t] += pds * hv[u][t];
        }
    }
    st.m = mc;
}

// normalise, bias, activation (layers.py:35,46) and the training extras
template <int FP, bool TRAIN>
__device__ __forceinline__ void write_row(const FwdArgs &a, const int64_t row, const RowState<TRAIN> &st,
                                          const int q, const int head, const float4_t &c4, const bool writer) {
    constexpr int K = HAN_D / FP;
    const float inv = st.l > 0.f ? 1.f / st.l : 0.f;
    // acc / accp carry the zero-or-keep terms; the two 1/keep factors go in here
    const float scale = TRAIN ? inv * a.inv_keep_coef * (a.lsb_mask ? a.inv_keep_fts : 1.f) : inv;
    float4_t pv, ov;
    float4_t r4 = {0.f, 0.f, 0.f, 0.f};
    if (a.res && writer) r4 = *reinterpret_cast<const float4_t *>(a.res + row * HAN_D + 4 * q);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        pv[t] = st.acc[t] * scale + c4[t] + r4[t];
        ov[t] = a.activation == HAN_ACT_ELU ? han_elu(pv[t]) : pv[t];
    }
    if (writer) {
        *reinterpret_cast<float4_t *>(a.out + row * a.out_stride + 4 * q) = ov;
        if (TRAIN) {
            float4_t ap;
#pragma unroll
            for (int t = 0; t < 4; ++t) ap[t] = st.accp[t] * scale;
            if (a.pre) *reinterpret_cast<float4_t *>(a.pre + row * HAN_D + 4 * q) = pv;      // optional: the backward reads `out`
            *reinterpret_cast<float4_t *>(a.aggp + row * HAN_D + 4 * q) = ap;
            if ((4 * q) % FP == 0) {
                a.lse[row * K + head] = st.l > 0.f ? st.m + __logf(st.l) : HAN_NEG_BIG;
                a.tsum[row * K + head] = st.tl * inv;
            }
        }
    }
}

// One wave per destination row (RPW = 1) or one 16-lane group per row (RPW = 4,
// for low-degree graphs).  TRAIN also produces pre / lse / aggp / tsum and
// applies the two dropouts.
// (Forcing more waves per SIMD on the TRAIN instantiation with __launch_bounds__ spills
// 92-180 B/lane to scratch and measured 5-25 % slower in both cache and HBM regimes.)
// SPLIT: a row's edges run as full U-steps + single steps for the tail (fewer VALU instructions;
// what the VALU-bound instantiations want) instead of masked U-steps (fewer registers: the fp32
// eval forward keeps 64 VGPRs = 8 waves/SIMD, which is what the HBM-bound regime wants)
template <int FP, bool TRAIN, int RPW, int U, bool BF, bool VAL, bool FAST, bool SPLIT, bool DD = false>
__global__ __launch_bounds__(256) void node_attn_fwd_kernel(const FwdArgs a_in) {
    FwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int K = HAN_D / FP;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int head = (4 * q) / FP;
    const int64_t wave0 = xcd_first_unit(threadIdx.x >> 6, a.xcd_order);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float4_t c4 = *reinterpret_cast<const float4_t *>(a.c + 4 * q);
    const float4_t a24 = *reinterpret_cast<const float4_t *>(a.a2 + 4 * q);
    const float b2h = a.b2[head];
    const bool drop_c = TRAIN && a.thr_coef < HAN_KEEP_ALL;
    const int64_t nunits = (a.n_work + RPW - 1) / RPW;   // wave-sized work units

    for (int64_t unit = wave0; unit < nunits; unit += nwaves) {
        const int64_t idx_raw = (RPW == 1) ? unit : unit * 4 + g;
        const bool row_ok = idx_raw < a.n_work;
        const int64_t widx = row_ok ? idx_raw : a.n_work - 1;
        const int64_t row = a.rows ? (int64_t)a.rows[widx] : widx;
        const int64_t s = a.rowptr[row];
        const int64_t e_raw = row_ok ? a.rowptr[row + 1] : s;
        const bool is_long = e_raw - s > a.split_deg;      // handled by the chunk kernels
        const int64_t e = is_long ? s : e_raw;
        const float f1h = a.f1[row * K + head];
        const uint32_t gi = (uint32_t)(row + a.row_offset);
        RowState<TRAIN> st;
        st.init();

        if (RPW == 1) {
            for (int64_t base = s; base < e; base += 64) {
                const int cnt = (int)((e - base) < 64 ? (e - base) : 64);
                const int mycol = a.colidx[base + (lane < cnt ? lane : cnt - 1)];
                const float myval = VAL ? a.edge_val[base + (lane < cnt ? lane : cnt - 1)] : 1.f;
                int it = 0;
                if constexpr (!SPLIT) {
                    for (; it * 4 < cnt; it += U) {          // masked steps
                        int j[U];
                        float w[U];
                        bool valid[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int idx = (it + u) * 4 + g;
                            valid[u] = idx < cnt;
                            j[u] = __shfl(mycol, idx & 63, 64);
                            w[u] = VAL ? __shfl(myval, idx & 63, 64) : 1.f;
                        }
                        consume_edges<FP, TRAIN, U, BF, VAL, FAST, false>(a, j, w, valid, f1h, gi, q, head, a24,
                                                                          b2h, drop_c, st);
                    }
                }
                for (; (it + U) * 4 <= cnt; it += U) {       // full steps: every slot is an edge
                    int j[U];
                    float w[U];
                    bool valid[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int idx = (it + u) * 4 + g;
                        valid[u] = true;
                        j[u] = __shfl(mycol, idx, 64);
                        w[u] = VAL ? __shfl(myval, idx, 64) : 1.f;
                    }
                    consume_edges<FP, TRAIN, U, BF, VAL, FAST, true, DD>(a, j, w, valid, f1h, gi, q, head, a24, b2h,
                                                                         drop_c, st);
                }
                if (U > 4 && (it + 4) * 4 <= cnt) {          // long unrolls: one half step before the singles
                    int j[4];
                    float w[4];
                    bool valid[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = (it + u) * 4 + g;
                        valid[u] = true;
                        j[u] = __shfl(mycol, idx, 64);
                        w[u] = VAL ? __shfl(myval, idx, 64) : 1.f;
                    }
                    consume_edges<FP, TRAIN, 4, BF, VAL, FAST, true>(a, j, w, valid, f1h, gi, q, head, a24, b2h,
                                                                     drop_c, st);
                    it += 4;
                }
                for (; it * 4 < cnt; ++it) {                 // tail: single steps of 4 edges
                    const int idx = it * 4 + g;
                    const int j[1] = {__shfl(mycol, idx & 63, 64)};
                    const float w[1] = {VAL ? __shfl(myval, idx & 63, 64) : 1.f};
                    const bool valid[1] = {idx < cnt};
                    consume_edges<FP, TRAIN, 1, BF, VAL, FAST, false>(a, j, w, valid, f1h, gi, q, head, a24, b2h,
                                                                      drop_c, st);
                }
            }
            st.merge(16);
            st.merge(32);
        } else {
            // each 16-lane group walks its own row; the wave loops to the longest.  The ids of up to 16 entries of a
            // row are ONE coalesced load of its group (lane q: entry base + q) handed out by shuffle, so a short row
            // (the degree-binned launch sends rows below 16 entries here) has a single index load in front of its gathers
            const int64_t len = e - s;
            int64_t maxlen = len;
            {
                int64_t o = __shfl_xor(maxlen, 16, 64);
                maxlen = o > maxlen ? o : maxlen;
                o = __shfl_xor(maxlen, 32, 64);
                maxlen = o > maxlen ? o : maxlen;
            }
            for (int64_t base = 0; base < maxlen; base += 16) {
                const int64_t left = len - base;                      // entries of THIS group's row from base on (may be <= 0)
                const int64_t at = s + base + (q < left ? q : (left > 0 ? left - 1 : 0));
                const int mycol = (left > 0) ? a.colidx[at] : 0;
                const float myval = (VAL && left > 0) ? a.edge_val[at] : 1.f;
                const int steps = (int)((maxlen - base) < 16 ? (maxlen - base) : 16);
                for (int it = 0; it < steps; it += U) {
                    int j[U];
                    float w[U];
                    bool valid[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        valid[u] = it + u < left;
                        j[u] = __shfl(mycol, (lane & 48) + ((it + u) & 15), 64);
                        w[u] = VAL ? __shfl(myval, (lane & 48) + ((it + u) & 15), 64) : 1.f;
                    }
                    consume_edges<FP, TRAIN, U, BF, VAL, FAST, false>(a, j, w, valid, f1h, gi, q, head, a24, b2h, drop_c, st);
                }
            }
        }

        write_row<FP, TRAIN>(a, row, st, q, head, c4, row_ok && !is_long && (RPW == 4 || g == 0));
    }
}

// ---------------------------------------------------------------------------------------------
// Lean kernels for SMALL graphs with long rows (HAN_FLAG_LEAN; the reference's own data sets: a few thousand
// nodes, ACM PSP 24 % dense, DBLP APCPA / APTPA 30 % / 78 %).  There the gather kernels above are not bound by
// memory but by vector-instruction issue (profiles/r03_pmc_k2_small_dense.json: 73-85 % VALU-active, one wave per
// row and ~4000 waves in all), so these kernels spend fewer vector instructions per edge:
//   * the neighbour score f2_j is READ from the K1 table (a 4-byte gather next to the row: free while the table lives
//     in the L2s, a memory line per edge on a large table -- hence small graphs only) instead of a dot product + lane
//     reduction per edge, and the softmax runs in log2 units (v_exp_f32 directly);
//   * the attention-dropout hash is computed once per (edge, four heads) by one lane of the group and handed out by
//     ds_bpermute, not by every lane;
//   * the ids of a step are one LDS read of the wave's staged ids, loaded one 64-entry piece ahead;
//   * for the reference shape (8 heads x 8 columns) one lane owns a whole head of an edge (8 groups of 8 lanes, 16
//     edges per step): everything that exists once per (edge, head) is computed once, dot products are in-lane.
// Same arithmetic per edge as the gather kernels up to the order of the sums.  fp32 tables, table index == global id.
// (An LDS-tiled form -- 16 rows per block walking the table in 256-row tiles -- was built first and measured: staging
// alone changed nothing, and with the same lean step it lost to these kernels at every density)
// ---------------------------------------------------------------------------------------------
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float han_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// the 16-bit field `head & 3` of the hash that lane (group base + u + 4 * (head >> 2)) computed
// the same field from the hash words of lane `u` of this lane's DPP quad (quad_perm broadcasts: no LDS round trip).  For
// the one-lane-per-head kernels: lanes 4c .. 4c+3 of an 8-lane group are the heads of head quad c, and lane 4c + u
// (u = 0, 1) hashes edge u of the step for that quad.
__device__ __forceinline__ uint32_t quad_field(const uint32_t hx, const uint32_t hy, const int u, const int head) {
    const uint32_t x = quad_bcast(hx, u), y = quad_bcast(hy, u);
    const uint32_t wsel = (head & 2) ? y : x;
    return (head & 1) ? (wsel >> 16) : (wsel & 0xFFFFu);
}

__device__ __forceinline__ uint32_t lean_field(const uint32_t hx, const uint32_t hy, const int src_addr, const int head) {
    const uint32_t x = (uint32_t)__builtin_amdgcn_ds_bpermute(src_addr, (int)hx);
    const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute(src_addr, (int)hy);
    const uint32_t wsel = (head & 2) ? y : x;
    return (head & 1) ? (wsel >> 16) : (wsel & 0xFFFFu);
}

// one step of 16 edges in the 16-lane map (group g takes the entries 4 g .. 4 g + 3 of the step)
template <int FP, bool TRAIN, bool VAL, bool MASKED>
__device__ __forceinline__ void lean_fwd_step(const FwdArgs &a, const float *tileH, const float *tileF, const int4 jj,
                                              const float4_t wv4, const int r, const uint32_t hoff, const uint32_t foff,
                                              const int j0, const float f1s, const uint32_t gi, const int g, const int q,
                                              const int head, const bool drop_c, const int (&baddr)[4],
                                              float &m, float &l, float &tl, float (&acc)[4], float (&accp)[4]) {
    constexpr int K = HAN_D / FP;
    constexpr int KQ = (K + 3) / 4;
    int j[4] = {jj.x, jj.y, jj.z, jj.w};
    bool valid[4];
    float4_t hv[4];
    float f2v[4], ev[4], sg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        valid[u] = !MASKED || (4 * g + u) < r;
        if (MASKED) j[u] = valid[u] ? j[u] : j0;           // a slot past the piece reads a row that exists
        hv[u] = *reinterpret_cast<const float4_t *>(tileH + ((int64_t)j[u] * HAN_D + hoff));
        f2v[u] = tileF[(int64_t)j[u] * K + foff];
    }
    uint32_t hx = 0, hy = 0;
    if (TRAIN && drop_c) {      // one hash per (edge, four heads): lane q of the group takes edge q & 3, head quad (q >> 2) % KQ
        const int uu = q & 3;
        const int ju = uu == 0 ? j[0] : (uu == 1 ? j[1] : (uu == 2 ? j[2] : j[3]));
        const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF, gi,
                                        (uint32_t)ju * (uint32_t)KQ + (uint32_t)((q >> 2) % KQ));
        hx = rn.x;
        hy = rn.y;
    }
    float mc = m;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float x = __builtin_fmaf(f2v[u], kLog2e, f1s);      // (f1_i + f2_j) * log2 e
        if (VAL) x *= wv4[u];
        if (TRAIN && a.lsb_mask) {                          // layers.py:31-32, after the score was taken (it comes from the table)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int bits = __float_as_int(hv[u][t]);
                hv[u][t] = __int_as_float(bits & han_bit_mask<0>(bits));
            }
        }
        if (TRAIN) {
            sg[u] = x > 0.f ? 1.f : a.slope;
            if (VAL) sg[u] *= wv4[u];
        }
        ev[u] = fmaxf(x, a.slope * x);
        if (MASKED) ev[u] = valid[u] ? ev[u] : HAN_NEG_BIG;
        mc = fmaxf(mc, ev[u]);
    }
    const float sc = han_exp2(m - mc);
    l *= sc;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] *= sc;
    if (TRAIN) {
        tl *= sc;
#pragma unroll
        for (int t = 0; t < 4; ++t) accp[t] *= sc;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float p = han_exp2(ev[u] - mc);
        if (MASKED) p = valid[u] ? p : 0.f;
        l += p;
        float pd = p;
        if (TRAIN && drop_c) pd = lean_field(hx, hy, baddr[u], head) < a.thr_coef ? p : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += pd * hv[u][t];
        if (TRAIN) {
            tl += p * sg[u];
            const float pds = pd * sg[u];
#pragma unroll
            for (int t = 0; t < 4; ++t) accp[t] += pds * hv[u][t];
        }
    }
    m = mc;
}

// The lean forward, any head shape (16-lane map): one wave per row, rows and scores gathered from global memory,
// steps of 16 edges, ids one piece ahead; whole rows of any length, any id order.
template <int FP, bool TRAIN, bool VAL>
__global__ __launch_bounds__(256) void node_attn_fwd_lean_kernel(const FwdArgs a_in) {
    if (a_in.only_if && *a_in.only_if == 0) return;      // the dense path did the work
    FwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int K = HAN_D / FP;
    __shared__ __attribute__((aligned(16))) int colw_all[4 * 64];
    __shared__ __attribute__((aligned(16))) float valw_all[VAL ? 4 * 64 : 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int *colw = colw_all + wv * 64;
    float *valw = valw_all + (VAL ? wv * 64 : 0);
    const int g = lane >> 4, q = lane & 15;
    const int head = (4 * q) / FP;
    const float4_t c4 = *reinterpret_cast<const float4_t *>(a.c + 4 * q);
    const bool drop_c = TRAIN && a.thr_coef < HAN_KEEP_ALL;
    const float *Hf = reinterpret_cast<const float *>(a.H);
    int baddr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) baddr[u] = (g * 16 + u + 4 * (head >> 2)) * 4;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < a.N; row += nwaves) {
        int64_t cur = a.rowptr[row];
        const int64_t e = a.rowptr[row + 1];
        const float f1s = a.f1[row * K + head] * kLog2e;
        const uint32_t gi = (uint32_t)(row + a.row_offset);
        float m = HAN_NEG_BIG, l = 0.f, tl = 0.f;
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, accp[4] = {0.f, 0.f, 0.f, 0.f};
        auto load_ids = [&](const int64_t at, int &col, float &val) {
            const int left = (int)((e - at) < 64 ? (e - at) : 64);
            if (left > 0) {
                col = a.colidx[at + (lane < left ? lane : left - 1)];
                if (VAL) val = a.edge_val[at + (lane < left ? lane : left - 1)];
            }
        };
        int nxt_col = 0;
        float nxt_val = 1.f;
        load_ids(cur, nxt_col, nxt_val);
        while (cur < e) {                                   // wave-uniform
            const int cnt = (int)((e - cur) < 64 ? (e - cur) : 64);
            const int mycol = nxt_col;
            const float myval = nxt_val;
            load_ids(cur + cnt, nxt_col, nxt_val);
            colw[lane] = mycol;
            if (VAL) valw[lane] = myval;
            const int nfull = cnt >> 4;
            int4 jj = *reinterpret_cast<const int4 *>(colw + 4 * g);
            float4_t w4 = {1.f, 1.f, 1.f, 1.f};
            if (VAL) w4 = *reinterpret_cast<const float4_t *>(valw + 4 * g);
            for (int it = 0; it < nfull; ++it) {
                const int nx = ((it + 1) & 3) * 16 + 4 * g;
                const int4 jn = *reinterpret_cast<const int4 *>(colw + nx);
                float4_t wn = {1.f, 1.f, 1.f, 1.f};
                if (VAL) wn = *reinterpret_cast<const float4_t *>(valw + nx);
                lean_fwd_step<FP, TRAIN, VAL, false>(a, Hf, a.f2g, jj, w4, 16, (uint32_t)(4 * q), (uint32_t)head, 0, f1s, gi, g,
                                                            q, head, drop_c, baddr, m, l, tl, acc, accp);
                jj = jn;
                w4 = wn;
            }
            if (nfull * 16 < cnt)
                lean_fwd_step<FP, TRAIN, VAL, true>(a, Hf, a.f2g, jj, w4, cnt - nfull * 16, (uint32_t)(4 * q), (uint32_t)head, 0,
                                                           f1s, gi, g, q, head, drop_c, baddr, m, l, tl, acc, accp);
            cur += cnt;
        }
        RowState<TRAIN> st;
        st.m = m * kLn2;
        st.l = l;
        st.tl = tl;
#pragma unroll
        for (int t = 0; t < 4; ++t) { st.acc[t] = acc[t]; st.accp[t] = accp[t]; }
        st.merge(16);
        st.merge(32);
        write_row<FP, TRAIN>(a, row, st, q, head, c4, g == 0);
    }
}

// ---------------------------------------------------------------------------------------------
// The lean forward with ONE LANE PER HEAD for the reference shape (8 heads x 8 columns; see
// node_attn_bwd_cols_h8_kernel): 8 groups of 8 lanes, lane h of a group owns the 8 columns of head h of one edge,
// 16 edges per step (two per group).  Score, LeakyReLU, exp, the dropout field and the softmax sums exist once per
// (edge, head) and are computed once.
// ---------------------------------------------------------------------------------------------
template <bool TRAIN, bool VAL>
__global__ __launch_bounds__(256) void node_attn_fwd_h8_kernel(const FwdArgs a_in) {
    if (a_in.only_if && *a_in.only_if == 0) return;      // the dense path did the work
    FwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    __shared__ __attribute__((aligned(16))) int colw_all[4 * 64];
    __shared__ __attribute__((aligned(16))) float valw_all[VAL ? 4 * 64 : 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int *colw = colw_all + wv * 64;
    float *valw = valw_all + (VAL ? wv * 64 : 0);
    const int g8 = lane >> 3, h = lane & 7;
    const bool drop_c = TRAIN && a.thr_coef < HAN_KEEP_ALL;
    const bool drop_f = TRAIN && a.lsb_mask;
    const float *Hf = reinterpret_cast<const float *>(a.H);
    const int pu = h & 1, pc = h >> 2;             // this lane hashes for (edge pu of its group, its own head quad): quad_field
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < a.N; row += nwaves) {
        int64_t cur = a.rowptr[row];
        const int64_t e = a.rowptr[row + 1];
        const float f1s = a.f1[row * 8 + h] * kLog2e;
        const uint32_t gi = (uint32_t)(row + a.row_offset);
        float m = HAN_NEG_BIG, l = 0.f, tl = 0.f;
        float acc[8], accp[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { acc[t] = 0.f; accp[t] = 0.f; }
        auto load_ids = [&](const int64_t at, int &col, float &val) {
            const int left = (int)((e - at) < 64 ? (e - at) : 64);
            if (left > 0) {
                col = a.colidx[at + (lane < left ? lane : left - 1)];
                if (VAL) val = a.edge_val[at + (lane < left ? lane : left - 1)];
            }
        };
        int nxt_col = 0;
        float nxt_val = 1.f;
        load_ids(cur, nxt_col, nxt_val);
        while (cur < e) {                                   // wave-uniform
            const int cnt = (int)((e - cur) < 64 ? (e - cur) : 64);
            colw[lane] = nxt_col;
            if (VAL) valw[lane] = nxt_val;
            load_ids(cur + cnt, nxt_col, nxt_val);
            for (int it = 0; it * 16 < cnt; ++it) {         // 16 edges per step: group g8 takes 16 it + 2 g8, + 1
                const int2 jj = *reinterpret_cast<const int2 *>(colw + it * 16 + 2 * g8);
                float2 ww = {1.f, 1.f};
                if (VAL) ww = *reinterpret_cast<const float2 *>(valw + it * 16 + 2 * g8);
                const int r = cnt - it * 16;
                int j[2] = {jj.x, jj.y};
                const float ew[2] = {ww.x, ww.y};
                bool valid[2];
                float hv[2][8], f2v[2], ev[2], sg[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    valid[u] = 2 * g8 + u < r;
                    j[u] = valid[u] ? j[u] : 0;
                    const float4_t h0 = *reinterpret_cast<const float4_t *>(Hf + (int64_t)j[u] * HAN_D + 8 * h);
                    const float4_t h1 = *reinterpret_cast<const float4_t *>(Hf + (int64_t)j[u] * HAN_D + 8 * h + 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { hv[u][t] = h0[t]; hv[u][4 + t] = h1[t]; }
                    f2v[u] = a.f2g[(int64_t)j[u] * 8 + h];
                }
                uint32_t hx = 0, hy = 0;
                if (drop_c) {
                    const int ju = pu ? j[1] : j[0];
                    const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF, gi, (uint32_t)ju * 2u + (uint32_t)pc);
                    hx = rn.x;
                    hy = rn.y;
                }
                float mc = m;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float x = __builtin_fmaf(f2v[u], kLog2e, f1s);      // (f1_i + f2_j) * log2 e
                    if (VAL) x *= ew[u];
                    if (TRAIN) {
                        sg[u] = x > 0.f ? 1.f : a.slope;
                        if (VAL) sg[u] *= ew[u];
                    }
                    if (drop_f) {                           // layers.py:31-32, after the score was taken (it comes from the table)
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const int bits = __float_as_int(hv[u][t]);
                            hv[u][t] = __int_as_float(bits & han_bit_mask<0>(bits));
                        }
                    }
                    ev[u] = valid[u] ? fmaxf(x, a.slope * x) : HAN_NEG_BIG;
                    mc = fmaxf(mc, ev[u]);
                }
                const float sc = han_exp2(m - mc);
                l *= sc;
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] *= sc;
                if (TRAIN) {
                    tl *= sc;
#pragma unroll
                    for (int t = 0; t < 8; ++t) accp[t] *= sc;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float p = valid[u] ? han_exp2(ev[u] - mc) : 0.f;
                    l += p;
                    float pd = p;
                    if (drop_c) pd = quad_field(hx, hy, u, h) < a.thr_coef ? p : 0.f;
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[t] += pd * hv[u][t];
                    if (TRAIN) {
                        tl += p * sg[u];
                        const float pds = pd * sg[u];
#pragma unroll
                        for (int t = 0; t < 8; ++t) accp[t] += pds * hv[u][t];
                    }
                }
                m = mc;
            }
            cur += cnt;
        }
        // merge the 8 groups (log2 units), then normalise, bias, activation (layers.py:35,46) and the training extras
#pragma unroll
        for (int off = 8; off <= 32; off <<= 1) {
            const float m_o = __shfl_xor(m, off, 64), l_o = __shfl_xor(l, off, 64);
            const float M = fmaxf(m, m_o);
            const float sa = han_exp2(m - M), sb = han_exp2(m_o - M);
            l = l * sa + l_o * sb;
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = acc[t] * sa + __shfl_xor(acc[t], off, 64) * sb;
            if (TRAIN) {
                tl = tl * sa + __shfl_xor(tl, off, 64) * sb;
#pragma unroll
                for (int t = 0; t < 8; ++t) accp[t] = accp[t] * sa + __shfl_xor(accp[t], off, 64) * sb;
            }
            m = M;
        }
        if (g8 == 0) {
            const float inv = l > 0.f ? 1.f / l : 0.f;
            const float scale = TRAIN ? inv * a.inv_keep_coef * (a.lsb_mask ? a.inv_keep_fts : 1.f) : inv;
            float4_t ov[2], ap[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const float4_t c4 = *reinterpret_cast<const float4_t *>(a.c + 8 * h + 4 * v);
                float4_t r4 = {0.f, 0.f, 0.f, 0.f};
                if (a.res) r4 = *reinterpret_cast<const float4_t *>(a.res + row * HAN_D + 8 * h + 4 * v);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float pv = acc[4 * v + t] * scale + c4[t] + r4[t];
                    ov[v][t] = a.activation == HAN_ACT_ELU ? han_elu(pv) : pv;
                    ap[v][t] = accp[4 * v + t] * scale;
                    if (TRAIN && a.pre) a.pre[row * HAN_D + 8 * h + 4 * v + t] = pv;
                }
                *reinterpret_cast<float4_t *>(a.out + row * a.out_stride + 8 * h + 4 * v) = ov[v];
                if (TRAIN) *reinterpret_cast<float4_t *>(a.aggp + row * HAN_D + 8 * h + 4 * v) = ap[v];
            }
            if (TRAIN) {
                a.lse[row * 8 + h] = l > 0.f ? m * kLn2 + __logf(l) : HAN_NEG_BIG;
                a.tsum[row * 8 + h] = tl * inv;
            }
        }
    }
}

// Split rows, step 1: one wave per chunk of a long row -> un-normalised partial state.
template <int FP, bool TRAIN, int U, bool BF, bool VAL>
__global__ __launch_bounds__(256) void node_attn_fwd_chunk_kernel(const FwdArgs a_in) {
    FwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int K = HAN_D / FP;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int head = (4 * q) / FP;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float4_t a24 = *reinterpret_cast<const float4_t *>(a.a2 + 4 * q);
    const float b2h = a.b2[head];
    const bool drop_c = TRAIN && a.thr_coef < HAN_KEEP_ALL;
    for (int64_t ch = wave0; ch < a.n_chunks; ch += nwaves) {
        const int64_t row = a.long_rows[a.chunk_long[ch]];
        const int64_t s = a.chunk_start[ch], e = a.chunk_end[ch];
        const float f1h = a.f1[row * K + head];
        const uint32_t gi = (uint32_t)(row + a.row_offset);
        RowState<TRAIN> st;
        st.init();
        for (int64_t base = s; base < e; base += 64) {
            const int cnt = (int)((e - base) < 64 ? (e - base) : 64);
            const int mycol = a.colidx[base + (lane < cnt ? lane : cnt - 1)];
            const float myval = VAL ? a.edge_val[base + (lane < cnt ? lane : cnt - 1)] : 1.f;
            for (int it = 0; it * 4 < cnt; it += U) {
                int j[U];
                float w[U];
                bool valid[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = (it + u) * 4 + g;
                    valid[u] = idx < cnt;
                    j[u] = __shfl(mycol, idx & 63, 64);
                    w[u] = VAL ? __shfl(myval, idx & 63, 64) : 1.f;
                }
                consume_edges<FP, TRAIN, U, BF, VAL, false, false>(a, j, w, valid, f1h, gi, q, head, a24, b2h, drop_c, st);
            }
        }
        st.merge(16);
        st.merge(32);
        if (g == 0) {
            float *w = a.split_ws + ch * kFwdChunkStride;
            float4_t v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = st.acc[t];
            *reinterpret_cast<float4_t *>(w + 4 * q) = v;
            if (TRAIN) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = st.accp[t];
                *reinterpret_cast<float4_t *>(w + 64 + 4 * q) = v;
            }
            if ((4 * q) % FP == 0) {
                w[128 + head] = st.m;
                w[128 + K + head] = st.l;
                if (TRAIN) w[128 + 2 * K + head] = st.tl;
            }
        }
    }
}

// Split rows, step 2: one BLOCK per long row merges its chunks.  (Rounds 1-3: one 16-lane group per row walked the
// row's chunks one after the other -- a chain of dependent loads: 100 us for the 245 chunks of a 10^6-edge row, 0.8 ms
// once rows are cut at 1024 entries, profiles/r04_k2_skew_split_sweep.jsonl.)  The 16 groups of the block take the
// chunks c = first + grp, + 16, ... (their loads do not depend on the running state, four are in flight), then group 0
// merges the 16 partial states in group order: a fixed order, bitwise reproducible.
template <int FP, bool TRAIN>
__global__ __launch_bounds__(256) void node_attn_fwd_finish_kernel(const FwdArgs a) {
    constexpr int K = HAN_D / FP;
    __shared__ float part[16][16][12];      // [group][lane q]: acc 4 | accp 4 | m | l | tl
    const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int head = (4 * q) / FP;
    const int64_t r = blockIdx.x;
    if (r >= a.n_long) return;
    const float4_t c4 = *reinterpret_cast<const float4_t *>(a.c + 4 * q);
    RowState<TRAIN> st;
    st.init();
    auto fold = [&](const float4_t &v, const float4_t &vp, const float m_o, const float l_o, const float tl_o) {
        const float M = fmaxf(st.m, m_o);
        const float sa = __expf(st.m - M), sb = __expf(m_o - M);
        st.l = st.l * sa + l_o * sb;
#pragma unroll
        for (int t = 0; t < 4; ++t) st.acc[t] = st.acc[t] * sa + v[t] * sb;
        if (TRAIN) {
            st.tl = st.tl * sa + tl_o * sb;
#pragma unroll
            for (int t = 0; t < 4; ++t) st.accp[t] = st.accp[t] * sa + vp[t] * sb;
        }
        st.m = M;
    };
    const int64_t c_end = a.long_ptr[r + 1];
    for (int64_t ch = a.long_ptr[r] + grp; ch < c_end; ch += 64) {
        float4_t v[4], vp[4];
        float mo[4], lo[4], to[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {        // four chunk states in flight
            const bool ok = ch + 16 * u < c_end;
            const float *w = a.split_ws + (ok ? ch + 16 * u : ch) * kFwdChunkStride;
            v[u] = *reinterpret_cast<const float4_t *>(w + 4 * q);
            vp[u] = TRAIN ? *reinterpret_cast<const float4_t *>(w + 64 + 4 * q) : v[u];
            mo[u] = ok ? w[128 + head] : HAN_NEG_BIG;
            lo[u] = ok ? w[128 + K + head] : 0.f;
            to[u] = (TRAIN && ok) ? w[128 + 2 * K + head] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ch + 16 * u < c_end) fold(v[u], vp[u], mo[u], lo[u], to[u]);      // uniform per 16-lane group
    }
    float *pp = part[grp][q];
#pragma unroll
    for (int t = 0; t < 4; ++t) { pp[t] = st.acc[t]; pp[4 + t] = TRAIN ? st.accp[t] : 0.f; }
    pp[8] = st.m; pp[9] = st.l; pp[10] = TRAIN ? st.tl : 0.f;
    __syncthreads();
    if (grp == 0) {
        st.init();
        for (int g = 0; g < 16; ++g) {
            const float *o = part[g][q];
            const float4_t v = {o[0], o[1], o[2], o[3]}, vp = {o[4], o[5], o[6], o[7]};
            fold(v, vp, o[8], o[9], o[10]);
        }
        write_row<FP, TRAIN>(a, a.long_rows[r], st, q, head, c4, true);
    }
}

// ---------------------------------------------------------------------------
// backward step 1: row-local pass
// ---------------------------------------------------------------------------
struct BwdRowsArgs {
    const float *dOut;
    int64_t dout_stride;
    const float *out;      // the forward's OUTPUT rows (after the activation): ELU is inverted, the pre-activation is not stored
    int64_t out_stride;
    const float *aggp, *tsum, *f1, *lse, *c;
    const float *res;   // residual term that was added to pre (or null)
    void *gs;      // fused [g | stats] rows (GsRow), g in fp32 or bf16
    float *df1;
    float *slab;   // [gridDim.x][64] partial sums of g (for dc)
    int64_t N;
    int activation;
};

template <int FP, bool BF>
__global__ __launch_bounds__(256) void node_attn_bwd_rows_kernel(const BwdRowsArgs a) {
    constexpr int K = HAN_D / FP;
    const int q = threadIdx.x & 15;
    const int head = (4 * q) / FP;
    const int64_t grp0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t ngrp = (int64_t)gridDim.x * 16;
    const float4_t c4 = *reinterpret_cast<const float4_t *>(a.c + 4 * q);
    float dc[4] = {0.f, 0.f, 0.f, 0.f};
    // all 16 lanes of a group run the same trip count -> the in-head sums are safe
    for (int64_t row = grp0; row < a.N; row += ngrp) {
        const float4_t d4 = *reinterpret_cast<const float4_t *>(a.dOut + row * a.dout_stride + 4 * q);
        const float4_t o4 = *reinterpret_cast<const float4_t *>(a.out + row * a.out_stride + 4 * q);
        const float4_t ap4 = *reinterpret_cast<const float4_t *>(a.aggp + row * HAN_D + 4 * q);
        float4_t g4;
        float4_t r4 = {0.f, 0.f, 0.f, 0.f};
        if (a.res) r4 = *reinterpret_cast<const float4_t *>(a.res + row * HAN_D + 4 * q);
        float sp = 0.f, dp = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // ELU'(pre) and pre from the output: out > 0: 1, out;  out <= 0: out + 1 (= e^pre), log(out + 1).  An output
            // that rounded to -1 has derivative 0; its pre-activation (below -17) is then irrelevant: any finite value
            const bool neg = a.activation == HAN_ACT_ELU && o4[t] <= 0.f;
            const float da = neg ? o4[t] + 1.f : 1.f;
            const float pt = neg ? (da > 0.f ? __logf(da) : 0.f) : o4[t];
            g4[t] = d4[t] * da;
            dc[t] += g4[t];                             // dc = sum_i g_i: row-local, exact g
            // bf16 g table: every consumer (the transposed-graph pass) sees the ROUNDED g, so the
            // row-local sums s_i and df1_i are formed from the rounded value too -- otherwise
            // sum_j dl_ij (gathered side) and df1_i (this side) disagree by the rounding of g
            if (BF) g4[t] = __uint_as_float(han_f32_to_bf16_bits(g4[t]) << 16);
            sp += g4[t] * (pt - c4[t] - r4[t]);         // g . (the aggregate alone)
            dp += g4[t] * ap4[t];
        }
        sp = head_sum<FP>(sp);
        dp = head_sum<FP>(dp);
        char *grow = reinterpret_cast<char *>(a.gs) + row * GsRow<FP, BF>::bytes;
        if (BF) {
            uint2 w;
            w.x = (__float_as_uint(g4[0]) >> 16) | (__float_as_uint(g4[1]) & 0xFFFF0000u);   // already rounded
            w.y = (__float_as_uint(g4[2]) >> 16) | (__float_as_uint(g4[3]) & 0xFFFF0000u);
            *reinterpret_cast<uint2 *>(grow + 8 * q) = w;
        } else {
            *reinterpret_cast<float4_t *>(grow + 16 * q) = g4;
        }
        if ((4 * q) % FP == 0) {
            const float ts = a.tsum[row * K + head];
            a.df1[row * K + head] = dp - sp * ts;
            float4_t st;
            st[0] = a.f1[row * K + head];
            st[1] = a.lse[row * K + head];
            st[2] = sp;
            st[3] = 0.f;
            *reinterpret_cast<float4_t *>(grow + GsRow<FP, BF>::g_bytes + 16 * head) = st;
        }
    }
    // block reduction of dc over the 16 groups
    __shared__ float red[16][64];
#pragma unroll
    for (int t = 0; t < 4; ++t) red[threadIdx.x >> 4][4 * q + t] = dc[t];
    __syncthreads();
    if (threadIdx.x < 64) {
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += red[r][threadIdx.x];
        a.slab[(int64_t)blockIdx.x * 64 + threadIdx.x] = sacc;
    }
}

// ---------------------------------------------------------------------------
// backward step 2: gather over the transposed graph (one wave per source row)
// ---------------------------------------------------------------------------
struct BwdColsArgs {
    const int64_t *colptr;
    const int32_t *rowidx;
    const float *edge_val;   // adjacency values in transposed-graph order, or null (binary)
    const void *gs, *H;   // fused [g | stats] rows of the destinations (GsRow); H: fp32 or bf16 local rows
    const int32_t *gid;  // global id of each row of the gs table, or null
    const float *f2, *df1, *a1, *a2;
    int lsb_mask;
    float *dH, *df2;
    int64_t NS;
    const int32_t *rows;     // degree-binned launches: the n_work source rows of this launch (null: 0 .. n_work-1)
    int64_t n_work;
    float slope;
    uint32_t seed_lo, seed_hi, thr_coef;
    const uint64_t *seed_dev;
    float inv_keep_coef, inv_keep_fts;
    int64_t src_offset, dst_offset;
    int xcd_order;
    int masked;          // HAN_FLAG_MASKED_EDGES
    int lean;            // HAN_FLAG_LEAN
    const int *only_if;  // lean kernel behind the dense path: run only when this device word is non-zero (null: always)
    int64_t split_deg;
    int64_t n_long, n_chunks;
    const int64_t *long_rows, *long_ptr, *chunk_start, *chunk_end;
    const int32_t *chunk_long;
    float *split_ws;
};

// per source row: the dropped projected row H~_j (layers.py:32) and its mask/keep factors
struct SrcRow {
    float4_t hd;
    float mk[4];
    float f2h;
    uint32_t gj;
};

template <int FP, bool BF>
__device__ __forceinline__ SrcRow load_src(const BwdColsArgs &a, const int64_t src, const int q, const int head) {
    constexpr int K = HAN_D / FP;
    SrcRow r;
    r.gj = (uint32_t)(src + a.src_offset);
    r.f2h = a.f2[src * K + head];
    r.hd = han_load_row4<BF>(a.H, src, q);
#pragma unroll
    for (int t = 0; t < 4; ++t) r.mk[t] = 1.f;
    if (a.lsb_mask) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            r.mk[t] = han_keep_bit<BF>(r.hd[t]) ? a.inv_keep_fts : 0.f;
            r.hd[t] *= r.mk[t];
        }
    }
    return r;
}

// gather U destinations i of source j and accumulate  acc += alpha~ g_i,  df += dl_ij
// MASKED (HAN_FLAG_MASKED_EDGES): entries of rowidx below 0 are edges whose destination cannot contribute (its g row is
// identically zero: a destination outside the loss mask of a one-layer model); they keep their POSITION in the
// row -- so every remaining term is added by the same lane group in the same order as in the full pass and the
// sums are bit-identical -- but nothing is loaded for them.
// DEDUP (U = 4 steps of HAN_FLAG_LEAN launches: small graphs, where this pass is bound by vector-instruction issue):
// the attention-dropout hash of an edge serves four heads, so ONE lane of the 16-lane group computes it (lane q: edge
// q & 3, head quad (q >> 2) % KQ) and the others fetch their 16-bit field with ds_bpermute, instead of all 16 hashing.
template <int FP, int U, bool BF, bool VAL, bool FAST, bool ALLV, bool MASKED = false, bool DEDUP = false>
__device__ __forceinline__ void bwd_consume(const BwdColsArgs &a, const int (&i)[U], const float (&ew)[U],
                                            const bool (&valid)[U], const SrcRow &sr, const int q, const int head, const bool drop_c,
                                            float (&acc)[4], float &dfacc) {
    constexpr int K = HAN_D / FP;
    constexpr int KQ = (K + 3) / 4;
    static_assert(!DEDUP || (U == 4 && !MASKED), "the shared hash is built for full 4-edge steps");
    uint32_t hx = 0, hy = 0;
    int baddr0 = 0;
    if (DEDUP && (FAST || drop_c)) {
        const int uu = q & 3;
        const int iu = uu == 0 ? i[0] : (uu == 1 ? i[1] : (uu == 2 ? i[2] : i[U - 1]));
        const int64_t gid_i = ((!FAST && a.gid) ? (int64_t)a.gid[iu] : (int64_t)iu) + a.dst_offset;
        const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF, (uint32_t)gid_i,
                                        sr.gj * (uint32_t)KQ + (uint32_t)((q >> 2) % KQ));
        hx = rn.x;
        hy = rn.y;
        baddr0 = (int)(((__lane_id() & 48) + 4 * (head >> 2)) * 4);
    }
    float4_t gv[U], st[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (MASKED) {
            gv[u] = (float4_t){0.f, 0.f, 0.f, 0.f};
            st[u] = (float4_t){0.f, 0.f, 0.f, 0.f};
            if (valid[u]) {
                gv[u] = gs_load_g4<FP, BF>(a.gs, (int64_t)i[u], q);
                st[u] = gs_load_stats<FP, BF>(a.gs, (int64_t)i[u], head);
            }
        } else {
            gv[u] = gs_load_g4<FP, BF>(a.gs, (int64_t)i[u], q);
            st[u] = gs_load_stats<FP, BF>(a.gs, (int64_t)i[u], head);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float x = st[u][0] + sr.f2h;
        if (VAL) x *= ew[u];
        float sg = x > 0.f ? 1.f : a.slope;
        if (VAL) sg *= ew[u];
        float alpha = __expf(han_lrelu(x, a.slope) - st[u][1]);
        alpha = (ALLV || valid[u]) ? alpha : 0.f;
        float am = 1.f;
        if (DEDUP) {
            if (FAST || drop_c) am = lean_field(hx, hy, baddr0 + 4 * u, head) < a.thr_coef ? a.inv_keep_coef : 0.f;
        } else if (FAST || drop_c) {
            const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF,
                                            (uint32_t)(((!FAST && a.gid) ? (int64_t)a.gid[(MASKED && !valid[u]) ? 0 : i[u]] : (int64_t)i[u]) + a.dst_offset),
                                            sr.gj * (uint32_t)KQ + (uint32_t)(head >> 2));
            am = rn.field(head & 3) < a.thr_coef ? a.inv_keep_coef : 0.f;
        }
        const float dot = head_sum<FP>(dot4(gv[u], sr.hd));
        dfacc += alpha * sg * (am * dot - st[u][2]);
        const float w = alpha * am;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += w * gv[u][t];
    }
}

template <int FP>
__device__ __forceinline__ void write_src(const BwdColsArgs &a, const int64_t src, const SrcRow &sr,
                                          const float (&acc)[4], const float dfacc, const int q, const int head,
                                          const float4_t &a14, const float4_t &a24) {
    constexpr int K = HAN_D / FP;
    const float d1 = a.df1[src * K + head];
    float4_t o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = acc[t] * sr.mk[t] + d1 * a14[t] + dfacc * a24[t];
    *reinterpret_cast<float4_t *>(a.dH + src * HAN_D + 4 * q) = o;
    if ((4 * q) % FP == 0) a.df2[src * K + head] = dfacc;
}

template <int FP, int RPW, int U, bool BF, bool VAL, bool FAST, bool MASKED = false, bool DEDUP = false>
__global__ __launch_bounds__(256) void node_attn_bwd_cols_kernel(const BwdColsArgs a_in) {
    BwdColsArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int head = (4 * q) / FP;
    const int64_t wave0 = xcd_first_unit(threadIdx.x >> 6, a.xcd_order);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const bool drop_c = a.thr_coef < HAN_KEEP_ALL;
    const float4_t a14 = *reinterpret_cast<const float4_t *>(a.a1 + 4 * q);
    const float4_t a24 = *reinterpret_cast<const float4_t *>(a.a2 + 4 * q);
    const int64_t nunits = (a.n_work + RPW - 1) / RPW;

    for (int64_t unit = wave0; unit < nunits; unit += nwaves) {
        const int64_t idx_raw = (RPW == 1) ? unit : unit * 4 + g;
        const bool src_ok = idx_raw < a.n_work;
        const int64_t widx = src_ok ? idx_raw : a.n_work - 1;
        const int64_t src = a.rows ? (int64_t)a.rows[widx] : widx;
        const int64_t s = a.colptr[src];
        const int64_t e_raw = src_ok ? a.colptr[src + 1] : s;
        const bool is_long = e_raw - s > a.split_deg;
        const int64_t e = is_long ? s : e_raw;
        const SrcRow sr = load_src<FP, BF>(a, src, q, head);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float dfacc = 0.f;
        const int64_t len = e - s;
        int64_t trips;
        if (RPW == 1) {
            trips = (len + 3) >> 2;
        } else {
            trips = len;
            int64_t o = __shfl_xor(trips, 16, 64);
            trips = o > trips ? o : trips;
            o = __shfl_xor(trips, 32, 64);
            trips = o > trips ? o : trips;
        }
        if (RPW == 1) {
            // as in the forward: the destination ids of 64 transposed edges are loaded coalesced
            // and handed out by shuffle, so no step (least of all a tail step) waits on an index load
            for (int64_t base = s; base < e; base += 64) {
                const int cnt = (int)((e - base) < 64 ? (e - base) : 64);
                const int myrow = a.rowidx[base + (lane < cnt ? lane : cnt - 1)];
                const float myval = VAL ? a.edge_val[base + (lane < cnt ? lane : cnt - 1)] : 1.f;
                int st = 0;
                for (; (st + U) * 4 <= cnt; st += U) {          // full steps: every slot is an edge
                    int i[U];
                    float ew[U];
                    bool valid[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int idx = (st + u) * 4 + g;
                        i[u] = __shfl(myrow, idx, 64);
                        valid[u] = MASKED ? i[u] >= 0 : true;
                        ew[u] = VAL ? __shfl(myval, idx, 64) : 1.f;
                    }
                    bwd_consume<FP, U, BF, VAL, FAST, !MASKED, MASKED, DEDUP && U == 4>(a, i, ew, valid, sr, q, head, drop_c, acc, dfacc);
                }
                if (U > 4 && (st + 4) * 4 <= cnt) {             // long unrolls: one half step before the singles
                    int i[4];
                    float ew[4];
                    bool valid[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = (st + u) * 4 + g;
                        i[u] = __shfl(myrow, idx, 64);
                        valid[u] = MASKED ? i[u] >= 0 : true;
                        ew[u] = VAL ? __shfl(myval, idx, 64) : 1.f;
                    }
                    bwd_consume<FP, 4, BF, VAL, FAST, !MASKED, MASKED, DEDUP>(a, i, ew, valid, sr, q, head, drop_c, acc, dfacc);
                    st += 4;
                }
                for (; st * 4 < cnt; ++st) {                    // tail: single steps of 4 edges
                    const int idx = st * 4 + g;
                    const int i[1] = {__shfl(myrow, idx & 63, 64)};
                    const bool valid[1] = {idx < cnt && (!MASKED || i[0] >= 0)};
                    const float ew[1] = {VAL ? __shfl(myval, idx & 63, 64) : 1.f};
                    bwd_consume<FP, 1, BF, VAL, FAST, false, MASKED>(a, i, ew, valid, sr, q, head, drop_c, acc, dfacc);
                }
            }
        }
        if (RPW != 1) {
            // one 16-lane group per source row: up to 16 destination ids per group in ONE coalesced load, handed
            // out by shuffle (as the forward)
            for (int64_t base = 0; base < trips; base += 16) {
                const int64_t left = len - base;
                const int64_t at = s + base + (q < left ? q : (left > 0 ? left - 1 : 0));
                const int myrow = (left > 0) ? a.rowidx[at] : 0;
                const float myval = (VAL && left > 0) ? a.edge_val[at] : 1.f;
                const int steps = (int)((trips - base) < 16 ? (trips - base) : 16);
                for (int st = 0; st < steps; st += U) {
                    int i[U];
                    float ew[U];
                    bool valid[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        valid[u] = st + u < left;
                        i[u] = __shfl(myrow, (lane & 48) + ((st + u) & 15), 64);
                        if (MASKED) valid[u] = valid[u] && i[u] >= 0;
                        ew[u] = VAL ? __shfl(myval, (lane & 48) + ((st + u) & 15), 64) : 1.f;
                    }
                    bwd_consume<FP, U, BF, VAL, FAST, false, MASKED>(a, i, ew, valid, sr, q, head, drop_c, acc, dfacc);
                }
            }
        }
        if (RPW == 1) {
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                dfacc += __shfl_xor(dfacc, off, 64);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] += __shfl_xor(acc[t], off, 64);
            }
        }
        if (src_ok && !is_long && (RPW == 4 || g == 0)) write_src<FP>(a, src, sr, acc, dfacc, q, head, a14, a24);
    }
}

// ---------------------------------------------------------------------------------------------
// One lane per head (the reference shape, 8 heads x 8 columns) for small graphs (HAN_FLAG_LEAN): a wave is 8 groups
// of 8 lanes, lane h of a group owns ALL 8 columns of head h of one edge, so everything that exists once per
// (edge, head) -- score, LeakyReLU, exp, dropout field, the dl term -- is computed once instead of by the two lanes
// that share a head in the 16-lane map, the dot product g_i . H~_j is in-lane (no DPP), and a step covers 8 edges.
// 3x fewer vector instructions per edge in this pass, which on small graphs is bound by exactly that
// (profiles/r03_pmc_k2_small_dense.json).  Not for large tables: 118-164 registers for the same rows in flight lost
// to the 16-lane map there (the bf16 lane-map experiment).  fp32 tables, table index == global id.
// ---------------------------------------------------------------------------------------------
template <bool VAL>
__global__ __launch_bounds__(256) void node_attn_bwd_cols_h8_kernel(const BwdColsArgs a_in) {
    if (a_in.only_if && *a_in.only_if == 0) return;      // the dense path did the work
    BwdColsArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int RB = GsRow<8, false>::bytes;
    constexpr int GB = GsRow<8, false>::g_bytes;
    __shared__ __attribute__((aligned(16))) int colw_all[4 * 64];
    __shared__ __attribute__((aligned(16))) float valw_all[VAL ? 4 * 64 : 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int *colw = colw_all + wv * 64;
    float *valw = valw_all + (VAL ? wv * 64 : 0);
    const int g8 = lane >> 3, h = lane & 7;
    const bool drop_c = a.thr_coef < HAN_KEEP_ALL;
    const float *Hf = reinterpret_cast<const float *>(a.H);
    const char *gsb = reinterpret_cast<const char *>(a.gs);
    // the hash of (edge u, head quad c) is computed by lane 4c + u of the group (lanes 4c + 2, 4c + 3 duplicate them)
    const int pu = h & 1, pc = h >> 2;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t src = wave0; src < a.NS; src += nwaves) {
        int64_t cur = a.colptr[src];
        const int64_t e = a.colptr[src + 1];
        const uint32_t gj = (uint32_t)(src + a.src_offset);
        const float f2h = a.f2[src * 8 + h];
        float hd[8], mk[8];
        {
            const float4_t h0 = *reinterpret_cast<const float4_t *>(Hf + src * HAN_D + 8 * h);
            const float4_t h1 = *reinterpret_cast<const float4_t *>(Hf + src * HAN_D + 8 * h + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { hd[t] = h0[t]; hd[4 + t] = h1[t]; }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                mk[t] = 1.f;
                if (a.lsb_mask) {
                    mk[t] = han_keep_bit<false>(hd[t]) ? a.inv_keep_fts : 0.f;
                    hd[t] *= mk[t];
                }
            }
        }
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float dfacc = 0.f;
        auto load_ids = [&](const int64_t at, int &col, float &val) {
            const int left = (int)((e - at) < 64 ? (e - at) : 64);
            if (left > 0) {
                col = a.rowidx[at + (lane < left ? lane : left - 1)];
                if (VAL) val = a.edge_val[at + (lane < left ? lane : left - 1)];
            }
        };
        int nxt_row = 0;
        float nxt_val = 1.f;
        load_ids(cur, nxt_row, nxt_val);
        while (cur < e) {                                   // wave-uniform
            const int cnt = (int)((e - cur) < 64 ? (e - cur) : 64);
            colw[lane] = nxt_row;
            if (VAL) valw[lane] = nxt_val;
            load_ids(cur + cnt, nxt_row, nxt_val);
            for (int it = 0; it * 16 < cnt; ++it) {         // 16 edges per step: group g8 takes 16 it + 2 g8, + 1
                const int2 ii = *reinterpret_cast<const int2 *>(colw + it * 16 + 2 * g8);
                float2 ww = {1.f, 1.f};
                if (VAL) ww = *reinterpret_cast<const float2 *>(valw + it * 16 + 2 * g8);
                const int r = cnt - it * 16;                // edges left in this piece (>= 16: all slots are edges)
                int i[2] = {ii.x, ii.y};
                const float ew[2] = {ww.x, ww.y};
                bool valid[2];
                float4_t g0[2], g1[2], st[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    valid[u] = 2 * g8 + u < r;
                    i[u] = valid[u] ? i[u] : 0;
                    const char *rowp = gsb + (int64_t)i[u] * RB;
                    g0[u] = *reinterpret_cast<const float4_t *>(rowp + 32 * h);
                    g1[u] = *reinterpret_cast<const float4_t *>(rowp + 32 * h + 16);
                    st[u] = *reinterpret_cast<const float4_t *>(rowp + GB + 16 * h);
                }
                uint32_t hx = 0, hy = 0;
                if (drop_c) {
                    const int iu = pu ? i[1] : i[0];
                    const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF, (uint32_t)((int64_t)iu + a.dst_offset),
                                                    gj * 2u + (uint32_t)pc);
                    hx = rn.x;
                    hy = rn.y;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float x = st[u][0] + f2h;
                    if (VAL) x *= ew[u];
                    float sg = x > 0.f ? 1.f : a.slope;
                    if (VAL) sg *= ew[u];
                    float alpha = __expf(han_lrelu(x, a.slope) - st[u][1]);
                    alpha = valid[u] ? alpha : 0.f;
                    float am = 1.f;
                    if (drop_c) am = quad_field(hx, hy, u, h) < a.thr_coef ? a.inv_keep_coef : 0.f;
                    float dot = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) dot += g0[u][t] * hd[t] + g1[u][t] * hd[4 + t];
                    dfacc += alpha * sg * (am * dot - st[u][2]);
                    const float w = alpha * am;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[t] += w * g0[u][t];
                        acc[4 + t] += w * g1[u][t];
                    }
                }
            }
            cur += cnt;
        }
#pragma unroll
        for (int off = 8; off <= 32; off <<= 1) {
            dfacc += __shfl_xor(dfacc, off, 64);
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] += __shfl_xor(acc[t], off, 64);
        }
        if (g8 == 0) {
            const float d1 = a.df1[src * 8 + h];
            float4_t o0, o1;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                o0[t] = acc[t] * mk[t] + d1 * a.a1[8 * h + t] + dfacc * a.a2[8 * h + t];
                o1[t] = acc[4 + t] * mk[4 + t] + d1 * a.a1[8 * h + 4 + t] + dfacc * a.a2[8 * h + 4 + t];
            }
            *reinterpret_cast<float4_t *>(a.dH + src * HAN_D + 8 * h) = o0;
            *reinterpret_cast<float4_t *>(a.dH + src * HAN_D + 8 * h + 4) = o1;
            a.df2[src * 8 + h] = dfacc;
        }
    }
}

// Split source rows: one wave per chunk -> partial sums; one 16-lane group per long row adds them.
template <int FP, int U, bool BF, bool VAL>
__global__ __launch_bounds__(256) void node_attn_bwd_chunk_kernel(const BwdColsArgs a_in) {
    BwdColsArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int head = (4 * q) / FP;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const bool drop_c = a.thr_coef < HAN_KEEP_ALL;
    for (int64_t ch = wave0; ch < a.n_chunks; ch += nwaves) {
        const int64_t src = a.long_rows[a.chunk_long[ch]];
        const int64_t s = a.chunk_start[ch], len = a.chunk_end[ch] - s;
        const SrcRow sr = load_src<FP, BF>(a, src, q, head);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float dfacc = 0.f;
        const int64_t trips = (len + 3) >> 2;
        for (int64_t it = 0; it < trips; it += U) {
            int i[U];
            float ew[U];
            bool valid[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = (it + u) * 4 + g;
                valid[u] = k < len;
                i[u] = a.rowidx[valid[u] ? s + k : s];
                valid[u] = valid[u] && i[u] >= 0;          // masked edges (HAN_FLAG_MASKED_EDGES) are skipped
                ew[u] = VAL ? a.edge_val[valid[u] ? s + k : s] : 1.f;
            }
            bwd_consume<FP, U, BF, VAL, false, false, true>(a, i, ew, valid, sr, q, head, drop_c, acc, dfacc);
        }
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            dfacc += __shfl_xor(dfacc, off, 64);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += __shfl_xor(acc[t], off, 64);
        }
        if (g == 0) {
            float *w = a.split_ws + ch * kBwdChunkStride;
            float4_t v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = acc[t];
            *reinterpret_cast<float4_t *>(w + 4 * q) = v;
            if ((4 * q) % FP == 0) w[64 + head] = dfacc;
        }
    }
}

// one BLOCK per long source row: the 16 groups sum the chunks c = first + grp, + 16, ... (four loads in flight), group 0
// adds the 16 partial sums in group order (fixed order: bitwise reproducible) -- see node_attn_fwd_finish_kernel
template <int FP, bool BF>
__global__ __launch_bounds__(256) void node_attn_bwd_finish_kernel(const BwdColsArgs a) {
    __shared__ float part[16][16][5];
    const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int head = (4 * q) / FP;
    const int64_t r = blockIdx.x;
    if (r >= a.n_long) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float dfacc = 0.f;
    const int64_t c_end = a.long_ptr[r + 1];
    for (int64_t ch = a.long_ptr[r] + grp; ch < c_end; ch += 64) {
        float4_t v[4];
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = ch + 16 * u < c_end;
            const float *w = a.split_ws + (ok ? ch + 16 * u : ch) * kBwdChunkStride;
            v[u] = *reinterpret_cast<const float4_t *>(w + 4 * q);
            d[u] = w[64 + head];
            if (!ok) { v[u] = (float4_t){0.f, 0.f, 0.f, 0.f}; d[u] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += v[u][t];
            dfacc += d[u];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) part[grp][q][t] = acc[t];
    part[grp][q][4] = dfacc;
    __syncthreads();
    if (grp == 0) {
        const float4_t a14 = *reinterpret_cast<const float4_t *>(a.a1 + 4 * q);
        const float4_t a24 = *reinterpret_cast<const float4_t *>(a.a2 + 4 * q);
        const int64_t src = a.long_rows[r];
        const SrcRow sr = load_src<FP, BF>(a, src, q, head);
        float tot[4] = {0.f, 0.f, 0.f, 0.f};
        float dtot = 0.f;
        for (int g = 0; g < 16; ++g) {
#pragma unroll
            for (int t = 0; t < 4; ++t) tot[t] += part[g][q][t];
            dtot += part[g][q][4];
        }
        write_src<FP>(a, src, sr, tot, dtot, q, head, a14, a24);
    }
}

// ---------------------------------------------------------------------------
// backward step 3: score-parameter gradients (da1, da2, db1, db2)
// slab row layout: [0,64) da1, [64,128) da2, [128,128+K) db1, [128+K,128+2K) db2
// ---------------------------------------------------------------------------
template <int FP, bool BF>
__global__ __launch_bounds__(256) void score_param_bwd_kernel(const void *H, const float *df1,
                                                              const float *df2, float *slab, int64_t N) {
    constexpr int K = HAN_D / FP;
    constexpr int WIDTH = 128 + 2 * K;
    const int q = threadIdx.x & 15;
    const int head = (4 * q) / FP;
    const int64_t grp0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t ngrp = (int64_t)gridDim.x * 16;
    float d1[4] = {0, 0, 0, 0}, d2[4] = {0, 0, 0, 0};
    float s1 = 0.f, s2 = 0.f;
    for (int64_t row = grp0; row < N; row += ngrp) {
        const float4_t h4 = han_load_row4<BF>(H, row, q);
        const float x1 = df1[row * K + head], x2 = df2[row * K + head];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            d1[t] += x1 * h4[t];
            d2[t] += x2 * h4[t];
        }
        s1 += x1;
        s2 += x2;
    }
    __shared__ float red[16][WIDTH];
    const int r = threadIdx.x >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        red[r][4 * q + t] = d1[t];
        red[r][64 + 4 * q + t] = d2[t];
    }
    if ((4 * q) % FP == 0) {
        red[r][128 + head] = s1;
        red[r][128 + K + head] = s2;
    }
    __syncthreads();
    if (threadIdx.x < WIDTH) {
        float sacc = 0.f;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) sacc += red[rr][threadIdx.x];
        slab[(int64_t)blockIdx.x * WIDTH + threadIdx.x] = sacc;
    }
}

// ---------------------------------------------------------------------------
// Attention coefficients as data (attn_head(..., return_coef=True), layers.py:43-44;
// models/gat.py:143-172 averages them over the heads).  Diagnostic output, not on the
// training path: one wave per destination row, lanes stride over the row's stored
// entries; pass 1 = online (max, sum) per head, pass 2 = write.
// ---------------------------------------------------------------------------
struct CoefArgs {
    const int64_t *rowptr;
    const int32_t *colidx;
    const float *edge_val;
    const int32_t *gid;
    const float *f1, *f2;
    float *coef;     // (E,K), or (E) when mean_heads
    int64_t N;
    float slope;
    uint32_t seed_lo, seed_hi, thr_coef;
    const uint64_t *seed_dev;
    float inv_keep_coef;
    int64_t row_offset;
    int mean_heads;
};

template <int K>
__global__ __launch_bounds__(256) void node_attn_coef_kernel(const CoefArgs a_in) {
    CoefArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int KQ = (K + 3) / 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const bool drop_c = a.thr_coef < HAN_KEEP_ALL;
    for (int64_t row = wave0; row < a.N; row += nwaves) {
        const int64_t s = a.rowptr[row], e = a.rowptr[row + 1];
        float f1r[K], m[K], l[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            f1r[k] = a.f1[row * K + k];
            m[k] = HAN_NEG_BIG;
            l[k] = 0.f;
        }
        for (int64_t p = s + lane; p < e; p += 64) {
            const int64_t j = a.colidx[p];
            const float w = a.edge_val ? a.edge_val[p] : 1.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float x = han_lrelu(w * (f1r[k] + a.f2[j * K + k]), a.slope);
                const float M = fmaxf(m[k], x);
                l[k] = l[k] * __expf(m[k] - M) + __expf(x - M);
                m[k] = M;
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float m_o = __shfl_xor(m[k], off, 64), l_o = __shfl_xor(l[k], off, 64);
                const float M = fmaxf(m[k], m_o);
                l[k] = l[k] * __expf(m[k] - M) + l_o * __expf(m_o - M);
                m[k] = M;
            }
        }
        const uint32_t gi = (uint32_t)(row + a.row_offset);
        for (int64_t p = s + lane; p < e; p += 64) {
            const int64_t j = a.colidx[p];
            const float w = a.edge_val ? a.edge_val[p] : 1.f;
            const uint32_t gj = a.gid ? (uint32_t)a.gid[j] : (uint32_t)j;
            float mean = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float x = han_lrelu(w * (f1r[k] + a.f2[j * K + k]), a.slope);
                float c = __expf(x - m[k]) / l[k];                                   // layers.py:27
                if (drop_c) {                                                        // layers.py:29-30
                    const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF, gi,
                                                    gj * (uint32_t)KQ + (uint32_t)(k >> 2));
                    c = rn.field(k & 3) < a.thr_coef ? c * a.inv_keep_coef : 0.f;
                }
                if (a.mean_heads) mean += c;
                else a.coef[p * K + k] = c;
            }
            if (a.mean_heads) a.coef[p] = mean * (1.f / K);
        }
    }
}

#include "node_attn_dense.h"

// segments of the column tiles so that a launch has ~1024 blocks (4 per CU, one wave of each per SIMD; measured at the
// DBLP APTPA shape: 256 / 768 / 1024 blocks = 82 / 64 / 62 us for the eval launch)
static void dense_split(int64_t rows, int64_t n_table, int *S, int *tps, int *tiles) {
    const int64_t bx = (rows + kDenseRowsPerBlock - 1) / kDenseRowsPerBlock;
    const int t = (int)((n_table + kDenseTile - 1) / kDenseTile);
    int64_t want = (1024 + bx - 1) / (bx > 0 ? bx : 1);
    if (want < 1) want = 1;
    if (want > t) want = t > 0 ? t : 1;
    *tps = (int)((t + want - 1) / want);
    if (*tps < 1) *tps = 1;
    *S = (t + *tps - 1) / *tps;
    if (*S < 1) *S = 1;
    *tiles = t;
}

static size_t dense_workspace_bytes(int64_t rows, int64_t n_table, int row_width) {
    int S, tps, tiles;
    dense_split(rows, n_table, &S, &tps, &tiles);
    return (size_t)kDenseHdrFloats * sizeof(float) + (size_t)S * (size_t)rows * (size_t)row_width * sizeof(float);
}

static bool dense_geometry(int64_t rows, int64_t n_table, bool train, const han_dense_t *dn, DenseArgs *d) {
    const int rw = train ? DenseRowWidth<true>::value : DenseRowWidth<false>::value;
    if (!dn->workspace || dn->workspace_bytes < dense_workspace_bytes(rows, n_table, rw) || n_table <= 0 ||

        dn->ld_words < (n_table + 31) / 32 || ((uintptr_t)dn->workspace & 15))
        return false;
    dense_split(rows, n_table, &d->S, &d->tiles_per_seg, &d->tiles);
    d->bits = dn->bits; d->ldw = dn->ld_words; d->NT = n_table;
    d->hdr = (float *)dn->workspace;
    d->slab = (float *)dn->workspace + kDenseHdrFloats;
    return true;
}

// CSR -> adjacency bit mask (rows must be zero-filled by the caller's memset in front of it); a bit that is already
// set is a repeated entry: counted, because the bit-mask form cannot hold a multigraph term
__global__ __launch_bounds__(256) void csr_to_bitmask_kernel(const int64_t *rowptr, const int32_t *colidx, int64_t N,
                                                             int64_t n_table, uint32_t *bits, int64_t ldw, int *dups) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < N; row += nwaves) {
        const int64_t e = rowptr[row + 1];
        for (int64_t p = rowptr[row] + lane; p < e; p += 64) {
            const int c = colidx[p];
            if (c < 0 || c >= n_table) {      // a masked (-1) or foreign id has no bit: reported like a repeated entry, so that
                if (dups) atomicAdd(dups, 1);      // the caller keeps such a graph off the dense form
                continue;
            }
            const uint32_t bit = 1u << (c & 31);
            const uint32_t old = atomicOr(bits + row * ldw + (c >> 5), bit);
            if ((old & bit) && dups) atomicAdd(dups, 1);
        }
    }
}

constexpr int kReduceBlocks = 1024;

bool fp_supported(int K, int FP) {
    return K * FP == HAN_D && (FP == 4 || FP == 8 || FP == 16 || FP == 32 || FP == 64);
}

// Launch geometry: 256-thread blocks, 4 waves each; cap the grid and grid-stride.
int attn_grid(int64_t units) { return han_grid_for(units, 4, 256 * 8 * 4); }

// mean degree (E / N) below which a 16-lane group per row beats a wave per row
constexpr double kLowDegree = 12.0;

bool split_ok(const han_row_split_t *sp, int64_t n_rows) {
    if (!sp) return true;
    // bins: a listed bin needs its list unless it is every row of the launch (identity)
    if (sp->n_short < 0 || sp->n_mid < 0 || sp->n_short + sp->n_mid > n_rows) return false;
    if (sp->n_short > 0 && !sp->short_rows && sp->n_short != n_rows) return false;
    if (sp->n_mid > 0 && !sp->mid_rows && sp->n_mid != n_rows) return false;
    if (sp->n_long == 0) return true;
    return sp->n_long > 0 && sp->n_chunks >= sp->n_long && sp->split_deg > 0 && sp->long_rows && sp->long_ptr &&
           sp->chunk_long && sp->chunk_start && sp->chunk_end && sp->workspace &&
           sp->workspace_bytes >= (size_t)sp->n_chunks * kFwdChunkStride * sizeof(float);
}

}  // namespace

#define HAN_DISPATCH_FP(FPV, ...)                                \
    switch (FPV) {                                               \
        case 4: { constexpr int FPC = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int FPC = 8; __VA_ARGS__; } break;   \
        case 16: { constexpr int FPC = 16; __VA_ARGS__; } break; \
        case 32: { constexpr int FPC = 32; __VA_ARGS__; } break; \
        default: { constexpr int FPC = 64; __VA_ARGS__; } break; \
    }

// bf16 tables: every head shape (the configs[4] shape 8 x 8 is the tuned one)
#define HAN_BF16_OK(FPV) (true)

// One launch over the rows `a.rows[0 .. a.n_work)` (or 0 .. n_work-1): a 16-lane group per row (short) or a wave per row.
template <int FPC, bool BF, bool VAL>
static void launch_fwd_rows(const FwdArgs &a, bool train, bool short_rows, hipStream_t st) {
    // FAST: the training configuration every shipped script uses (both dropouts on, table
    // index == global id) gets an edge loop without uniform branches
    const bool fast = train && a.thr_coef < HAN_KEEP_ALL && a.lsb_mask && !a.gid && !a.f2g;
    if (short_rows) {
        const int grid = attn_grid((a.n_work + 3) / 4);
        if (train && fast) node_attn_fwd_kernel<FPC, true, 4, 4, BF, VAL, true, false><<<grid, 256, 0, st>>>(a);
        else if (train) node_attn_fwd_kernel<FPC, true, 4, 4, BF, VAL, false, false><<<grid, 256, 0, st>>>(a);
        else node_attn_fwd_kernel<FPC, false, 4, 4, BF, VAL, false, false><<<grid, 256, 0, st>>>(a);
    } else {
        const int grid = attn_grid(a.n_work);
        // steps of 4 x 4 rows for both table types.  (Rounds 2-3 ran the bf16 eval forward with 8 steps in flight, 114-125
        // registers, four waves per SIMD; re-measured in round 4 at N = 10M, profiles/r04_k2_bf16_in_flight_sweep.jsonl:
        // 4 steps / 76 registers / six waves 12.3 ms, 8 steps 13.3 ms, 16 steps 21.0 ms, 2 masked steps at eight waves
        // 13.6 ms -- what a CU keeps in flight is waves x steps, and registers spent on deeper unrolls cost more waves
        // than they add rows.)
        constexpr int UE = 4;
        // One attention-dropout hash per lane and 4-edge step -- consume_edges<..., DD> -- is bitwise the same and no faster.
        // With ds_bpermute (profiles/r04_k2_bf16_in_flight_sweep.jsonl): fp32 25.4 -> 24.0 ms at N = 10M on one box, 25.5 ->
        // 25.7 on another, bf16 18.0 -> 18.9 ms, 1.84 / 1.56 -> 1.85 / 1.61 ms at N = 1M.  With DPP quad broadcasts
        // (profiles/r04_k2_train_fwd_experiments.jsonl): fp32 24.78 -> 25.00 ms, bf16 18.35 -> 18.41 ms at N = 10M, 1.829 ->
        // 1.842 / 1.553 -> 1.579 ms at N = 1M -- eleven vector instructions fewer per edge and nothing gained, so vector
        // issue is not what this kernel waits for.  Nor is it the number of gathers in flight: requesting the rows of a
        // full step one step ahead (two steps in flight per wave, same occupancy for fp32: 121 registers) moved the fp32
        // kernels by < 0.5 % at either size and cost the bf16 training forward its fourth wave (138 registers: 18.7 -> 22.7
        // ms); not in the tree.  HAN_FLAG_K2_SHARED_HASH selects the shared hash for measurements, and its test pins that
        // the draws are the same.
        constexpr bool DD_OK = FPC == 8 && !VAL;
        if (DD_OK && train && fast && a.shared_hash) {
            if constexpr (DD_OK) node_attn_fwd_kernel<FPC, true, 1, 4, BF, VAL, true, true, true><<<grid, 256, 0, st>>>(a);
        } else if (train && fast) node_attn_fwd_kernel<FPC, true, 1, 4, BF, VAL, true, true><<<grid, 256, 0, st>>>(a);
        else if (train) node_attn_fwd_kernel<FPC, true, 1, 4, BF, VAL, false, true><<<grid, 256, 0, st>>>(a);
        else if (BF && a.deep) node_attn_fwd_kernel<FPC, false, 1, 8, BF, VAL, false, BF><<<grid, 256, 0, st>>>(a);      // HAN_FLAG_K2_DEEP
        else node_attn_fwd_kernel<FPC, false, 1, UE, BF, VAL, false, BF><<<grid, 256, 0, st>>>(a);
    }
}

// Degree bins of a launch (han_row_split_t): rows below HAN_SHORT_DEG entries, rows up to split_deg, rows beyond.
struct RowBins {
    bool binned;
    int64_t n_short, n_mid;
    const int32_t *short_rows, *mid_rows;
};

static RowBins bins_of(const han_row_split_t *sp) {
    RowBins b = {false, 0, 0, nullptr, nullptr};
    if (sp && (sp->n_short > 0 || sp->n_mid > 0)) {
        b.binned = true;
        b.n_short = sp->n_short; b.n_mid = sp->n_mid;
        b.short_rows = sp->short_rows; b.mid_rows = sp->mid_rows;
    }
    return b;
}

template <int FPC, bool BF, bool VAL>
static void launch_fwd_v(const FwdArgs &a_in, bool train, bool low, bool has_split, const RowBins &bins, hipStream_t st) {
    FwdArgs a = a_in;
    if (bins.binned) {
        // degree-binned: rows below 16 entries four to a wave, the others a wave each (rows beyond split_deg: chunks)
        if (bins.n_short > 0) {
            a.rows = bins.short_rows; a.n_work = bins.n_short;
            launch_fwd_rows<FPC, BF, VAL>(a, train, true, st);
        }
        if (bins.n_mid > 0) {
            a.rows = bins.mid_rows; a.n_work = bins.n_mid;
            launch_fwd_rows<FPC, BF, VAL>(a, train, false, st);
        }
    } else {
        a.rows = nullptr; a.n_work = a.N;
        launch_fwd_rows<FPC, BF, VAL>(a, train, low, st);
    }
    if (has_split) {
        const int cgrid = attn_grid(a.n_chunks);
        const int fgrid = (int)a.n_long;      // one block per long row
        if (train) {
            node_attn_fwd_chunk_kernel<FPC, true, 4, BF, VAL><<<cgrid, 256, 0, st>>>(a);
            node_attn_fwd_finish_kernel<FPC, true><<<fgrid, 256, 0, st>>>(a);
        } else {
            node_attn_fwd_chunk_kernel<FPC, false, 4, BF, VAL><<<cgrid, 256, 0, st>>>(a);
            node_attn_fwd_finish_kernel<FPC, false><<<fgrid, 256, 0, st>>>(a);
        }
    }
}

template <int FPC, bool BF, bool VAL>
static void launch_bwd_rows(const BwdColsArgs &a, bool short_rows, hipStream_t st) {
    const bool fast = a.thr_coef < HAN_KEEP_ALL && !a.gid;
    if (a.masked) {      // opt-in masked backward: the general instantiations, dead entries skipped in place
        if (short_rows) node_attn_bwd_cols_kernel<FPC, 4, 4, BF, VAL, false, true><<<attn_grid((a.n_work + 3) / 4), 256, 0, st>>>(a);
        else node_attn_bwd_cols_kernel<FPC, 1, 4, BF, VAL, false, true><<<attn_grid(a.n_work), 256, 0, st>>>(a);
    } else if (short_rows) {
        if (fast) node_attn_bwd_cols_kernel<FPC, 4, 4, BF, VAL, true><<<attn_grid((a.n_work + 3) / 4), 256, 0, st>>>(a);
        else node_attn_bwd_cols_kernel<FPC, 4, 4, BF, VAL, false><<<attn_grid((a.n_work + 3) / 4), 256, 0, st>>>(a);
    } else if (fast) node_attn_bwd_cols_kernel<FPC, 1, 4, BF, VAL, true><<<attn_grid(a.n_work), 256, 0, st>>>(a);
    else node_attn_bwd_cols_kernel<FPC, 1, 4, BF, VAL, false><<<attn_grid(a.n_work), 256, 0, st>>>(a);
}

template <int FPC, bool BF, bool VAL>
static void launch_bwd_cols_v(const BwdColsArgs &a_in, bool low, bool has_split, const RowBins &bins, hipStream_t st) {
    BwdColsArgs a = a_in;
    a.rows = nullptr; a.n_work = a.NS;
    if (a.lean && !a.masked && !low && !BF && FPC == 8 && !a.gid) {      // small graphs, 8 x 8: one lane per head
        node_attn_bwd_cols_h8_kernel<VAL><<<attn_grid(a.NS), 256, 0, st>>>(a);
        return;      // whole rows of any length: no chunk / finish launches behind it
    } else if (a.lean && !a.masked && !low && !BF) {      // small graphs (HAN_FLAG_LEAN): VALU-bound, one hash per (edge, four heads)
        a.split_deg = INT64_MAX;
        node_attn_bwd_cols_kernel<FPC, 1, 4, false, VAL, false, false, true><<<attn_grid(a.NS), 256, 0, st>>>(a);
        return;
    } else if (bins.binned) {
        if (bins.n_short > 0) {
            a.rows = bins.short_rows; a.n_work = bins.n_short;
            launch_bwd_rows<FPC, BF, VAL>(a, true, st);
        }
        if (bins.n_mid > 0) {
            a.rows = bins.mid_rows; a.n_work = bins.n_mid;
            launch_bwd_rows<FPC, BF, VAL>(a, false, st);
        }
    } else {
        launch_bwd_rows<FPC, BF, VAL>(a, low, st);
    }
    if (has_split) {
        node_attn_bwd_chunk_kernel<FPC, 4, BF, VAL><<<attn_grid(a.n_chunks), 256, 0, st>>>(a);
        node_attn_bwd_finish_kernel<FPC, BF><<<(int)a.n_long, 256, 0, st>>>(a);
    }
}

template <int FPC, bool VAL>
static void launch_fwd_lean_v(const FwdArgs &a, bool train, hipStream_t st) {
    const int grid = attn_grid(a.N);
    if (FPC == 8) {       // the reference shape: one lane per head
        if (train) node_attn_fwd_h8_kernel<true, VAL><<<grid, 256, 0, st>>>(a);
        else node_attn_fwd_h8_kernel<false, VAL><<<grid, 256, 0, st>>>(a);
    } else if (train) node_attn_fwd_lean_kernel<FPC, true, VAL><<<grid, 256, 0, st>>>(a);
    else node_attn_fwd_lean_kernel<FPC, false, VAL><<<grid, 256, 0, st>>>(a);
}

// the binary-adjacency instantiation (every shipped config) carries no edge-value registers
template <int FPC, bool BF>
static void launch_fwd(const FwdArgs &a, bool train, bool low, bool has_split, const RowBins &bins, hipStream_t st) {
    if (a.edge_val) launch_fwd_v<FPC, BF, true>(a, train, low, has_split, bins, st);
    else launch_fwd_v<FPC, BF, false>(a, train, low, has_split, bins, st);
}
template <int FPC, bool BF>
static void launch_bwd_cols(const BwdColsArgs &a, bool low, bool has_split, const RowBins &bins, hipStream_t st) {
    if (a.edge_val) launch_bwd_cols_v<FPC, BF, true>(a, low, has_split, bins, st);
    else launch_bwd_cols_v<FPC, BF, false>(a, low, has_split, bins, st);
}

static bool dtype_ok(int dt, int FP) {
    return dt == HAN_DTYPE_F32 || (dt == HAN_DTYPE_BF16 && HAN_BF16_OK(FP));
}

extern "C" int han_node_attn_fwd(const int64_t *rowptr, const int32_t *colidx, const float *edge_val,
                                 const void *H, int table_dtype,
                                 const int32_t *table_gid, const float *f1, const float *f2_src, const float *a2,
                                 const float *b2,
                                 const float *c, const float *res, float *out, int64_t out_stride, float *pre,
                                 float *lse,
                                 float *aggp, float *tsum, int64_t N, int64_t E, int K, int FP, float slope,
                                 float coef_drop, float fts_drop, uint64_t seed, const uint64_t *seed_dev,
                                 int64_t row_offset, int activation, int flags, const han_row_split_t *split,
                                 const han_dense_t *dense, void *stream) {
    if (N == 0) return 0;   // nothing to do; row pointers of empty tensors may be null
    if (!rowptr || (!colidx && E > 0) || !H || !f1 || !a2 || !b2 || !c || !out || N < 0 || E < 0 || out_stride < HAN_D)
        return HAN_E_BADARG;
    if (!split_ok(split, N)) return HAN_E_BADARG;
    if (!fp_supported(K, FP) || !dtype_ok(table_dtype, FP)) return HAN_E_UNSUPPORTED;
    if ((flags & HAN_FLAG_LEAN) && !f2_src) return HAN_E_BADARG;      // the lean kernels read the scores from the table
    const bool train = pre || lse || aggp || tsum;
    if (train && !(lse && aggp && tsum)) return HAN_E_BADARG;      // pre is optional: the backward works from `out`
    if (coef_drop < 0.f || coef_drop >= 1.f || fts_drop < 0.f || fts_drop >= 1.f) return HAN_E_BADARG;
    if ((coef_drop > 0.f || fts_drop > 0.f) && !train) return HAN_E_BADARG;
    if (N == 0) return 0;
    FwdArgs a;
    a.rowptr = rowptr; a.colidx = colidx; a.edge_val = edge_val; a.H = H; a.gid = table_gid; a.lsb_mask = fts_drop > 0.f; a.f1 = f1; a.f2g = f2_src; a.a2 = a2; a.b2 = b2; a.c = c; a.res = res;
    a.out = out; a.out_stride = out_stride; a.pre = pre; a.lse = lse; a.aggp = aggp; a.tsum = tsum;
    a.N = N; a.rows = nullptr; a.n_work = N; a.only_if = nullptr; a.deep = (flags & HAN_FLAG_K2_DEEP) ? 1 : 0; a.shared_hash = (flags & HAN_FLAG_K2_SHARED_HASH) ? 1 : 0; a.slope = slope;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_dev = seed_dev;
    a.thr_coef = coef_drop > 0.f ? han_keep_threshold(1.f - coef_drop) : HAN_KEEP_ALL;
    a.inv_keep_coef = 1.f / (1.f - coef_drop);
    a.inv_keep_fts = 1.f / (1.f - fts_drop);
    a.row_offset = row_offset; a.activation = activation; a.xcd_order = (flags & HAN_FLAG_XCD_ORDER) ? 1 : 0;
    const bool has_split = split && split->n_long > 0;
    a.split_deg = has_split ? split->split_deg : INT64_MAX;
    a.n_long = has_split ? split->n_long : 0;
    a.n_chunks = has_split ? split->n_chunks : 0;
    a.long_rows = has_split ? split->long_rows : nullptr;
    a.long_ptr = has_split ? split->long_ptr : nullptr;
    a.chunk_long = has_split ? split->chunk_long : nullptr;
    a.chunk_start = has_split ? split->chunk_start : nullptr;
    a.chunk_end = has_split ? split->chunk_end : nullptr;
    a.split_ws = has_split ? (float *)split->workspace : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const bool low = (double)E < kLowDegree * (double)N;
    if (dense && dense->bits) {
        // small dense graph on the matrix pipe (node_attn_dense.h); the lean CSR kernel behind it runs only when the
        // scores' range is too wide for the fixed-shift form (a device-side flag: no host round trip)
        if (!(flags & HAN_FLAG_LEAN) || table_dtype != HAN_DTYPE_F32 || table_gid || !f2_src || edge_val || K != 8 || FP != 8)
            return HAN_E_BADARG;
        DenseArgs d;
        if (!dense_geometry(N, dense->n_table, train, dense, &d)) return HAN_E_WORKSPACE;
        dense_f2_range_kernel<<<kDenseRangeBlocks, 256, 0, st>>>(f2_src, d.NT, d.hdr);
        const dim3 dg((unsigned)((N + kDenseRowsPerBlock - 1) / kDenseRowsPerBlock), (unsigned)d.S);
        const int fg = (int)((N + 15) / 16);
        if (train) {
            node_attn_fwd_dense_kernel<true><<<dg, 256, 0, st>>>(a, d);
            node_attn_fwd_dense_finish_kernel<true><<<fg, 256, 0, st>>>(a, d);
        } else {
            node_attn_fwd_dense_kernel<false><<<dg, 256, 0, st>>>(a, d);
            node_attn_fwd_dense_finish_kernel<false><<<fg, 256, 0, st>>>(a, d);
        }
        HAN_CHECK_LAUNCH();
        a.only_if = reinterpret_cast<const int *>(d.hdr) + 16;
    }
    if ((flags & HAN_FLAG_LEAN) && table_dtype == HAN_DTYPE_F32 && !table_gid && f2_src) {
        // small graph, table in the L2s: scores gathered, one hash per (edge, four heads); whole rows (no row split)
        if (edge_val) { HAN_DISPATCH_FP(FP, { launch_fwd_lean_v<FPC, true>(a, train, st); }) }
        else { HAN_DISPATCH_FP(FP, { launch_fwd_lean_v<FPC, false>(a, train, st); }) }
        HAN_CHECK_LAUNCH();
        return 0;
    }
    const RowBins bins = bins_of(split);
    if (table_dtype == HAN_DTYPE_BF16) {
        HAN_DISPATCH_FP(FP, { launch_fwd<FPC, true>(a, train, low, has_split, bins, st); })
    } else {
        HAN_DISPATCH_FP(FP, { launch_fwd<FPC, false>(a, train, low, has_split, bins, st); })
    }
    HAN_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t han_row_split_workspace(int64_t n_chunks) {
    return (size_t)(n_chunks > 0 ? n_chunks : 0) * kFwdChunkStride * sizeof(float);
}

extern "C" size_t han_gs_row_bytes(int K, int FP, int table_dtype) {
    if (!fp_supported(K, FP) || !dtype_ok(table_dtype, FP)) return 0;
    const size_t g_bytes = (size_t)HAN_D * (table_dtype == HAN_DTYPE_BF16 ? 2 : 4);
    return ((g_bytes + 16 * (size_t)K + 127) / 128) * 128;
}

extern "C" size_t han_node_attn_bwd_workspace(int64_t N, int K, int FP) {
    (void)N; (void)K; (void)FP;
    return (size_t)kReduceBlocks * 64 * sizeof(float);
}

extern "C" int han_node_attn_bwd_rows(const float *dOut, int64_t dout_stride, const float *out, int64_t out_stride,
                                      const float *aggp, const float *tsum, const float *f1,
                                      const float *lse, const float *c, const float *res, void *gs,
                                      int table_dtype, float *df1, float *dc, void *workspace,
                                      size_t workspace_bytes, int64_t N, int K, int FP, int activation,
                                      void *stream) {
    if (!dOut || !out || !aggp || !tsum || !f1 || !lse || !c || !gs || !df1 || !dc || !workspace ||
        N < 0 || dout_stride < HAN_D || out_stride < HAN_D)
        return HAN_E_BADARG;
    if (!fp_supported(K, FP) || !dtype_ok(table_dtype, FP)) return HAN_E_UNSUPPORTED;
    if (workspace_bytes < han_node_attn_bwd_workspace(N, K, FP)) return HAN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    BwdRowsArgs a;
    a.dOut = dOut; a.dout_stride = dout_stride; a.out = out; a.out_stride = out_stride; a.aggp = aggp; a.tsum = tsum;
    a.f1 = f1; a.lse = lse; a.c = c; a.res = res; a.gs = gs; a.df1 = df1;
    a.slab = (float *)workspace; a.N = N; a.activation = activation;
    const int grid = han_grid_for(N, 16, kReduceBlocks);
    if (table_dtype == HAN_DTYPE_BF16) {
        HAN_DISPATCH_FP(FP, { node_attn_bwd_rows_kernel<FPC, true><<<grid, 256, 0, st>>>(a); })
    } else {
        HAN_DISPATCH_FP(FP, { node_attn_bwd_rows_kernel<FPC, false><<<grid, 256, 0, st>>>(a); })
    }
    HAN_CHECK_LAUNCH();
    hipError_t e = han_reduce_slabs((const float *)workspace, grid, 64, 64, han_reduce_to(dc, 64), st);
    if (e != hipSuccess) return (int)e;
    return 0;
}

extern "C" int han_node_attn_bwd_cols(const int64_t *colptr, const int32_t *rowidx, const float *edge_val,
                                      const void *gs, const int32_t *table_gid, const void *H,
                                      int table_dtype, const float *f2,
                                      const float *df1, const float *a1, const float *a2,
                                      float *dH, float *df2, int64_t NS, int64_t E, int K, int FP, float slope,
                                      float coef_drop, float fts_drop, uint64_t seed, const uint64_t *seed_dev,
                                      int64_t src_offset, int64_t dst_offset, int flags,
                                      const han_row_split_t *split, const han_dense_t *dense, void *stream) {
    if (!colptr || (!rowidx && E > 0) || !gs || !H || !f2 || !df1 || !a1 || !a2 || !dH || !df2 || NS < 0 || E < 0)
        return HAN_E_BADARG;
    if (!split_ok(split, NS)) return HAN_E_BADARG;
    if (!fp_supported(K, FP) || !dtype_ok(table_dtype, FP)) return HAN_E_UNSUPPORTED;
    if (coef_drop < 0.f || coef_drop >= 1.f || fts_drop < 0.f || fts_drop >= 1.f) return HAN_E_BADARG;
    if (NS == 0) return 0;
    BwdColsArgs a;
    a.colptr = colptr; a.rowidx = rowidx; a.edge_val = edge_val; a.gs = gs; a.gid = table_gid; a.H = H; a.lsb_mask = fts_drop > 0.f; a.f2 = f2;
    a.df1 = df1; a.a1 = a1; a.a2 = a2; a.dH = dH; a.df2 = df2; a.NS = NS; a.rows = nullptr; a.n_work = NS; a.slope = slope;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_dev = seed_dev;
    a.thr_coef = coef_drop > 0.f ? han_keep_threshold(1.f - coef_drop) : HAN_KEEP_ALL;
    a.inv_keep_coef = 1.f / (1.f - coef_drop);
    a.inv_keep_fts = 1.f / (1.f - fts_drop);
    a.src_offset = src_offset; a.dst_offset = dst_offset; a.xcd_order = (flags & HAN_FLAG_XCD_ORDER) ? 1 : 0;
    a.masked = (flags & HAN_FLAG_MASKED_EDGES) ? 1 : 0;
    a.lean = (flags & HAN_FLAG_LEAN) ? 1 : 0;
    a.only_if = nullptr;
    const bool has_split = split && split->n_long > 0;
    a.split_deg = has_split ? split->split_deg : INT64_MAX;
    a.n_long = has_split ? split->n_long : 0;
    a.n_chunks = has_split ? split->n_chunks : 0;
    a.long_rows = has_split ? split->long_rows : nullptr;
    a.long_ptr = has_split ? split->long_ptr : nullptr;
    a.chunk_long = has_split ? split->chunk_long : nullptr;
    a.chunk_start = has_split ? split->chunk_start : nullptr;
    a.chunk_end = has_split ? split->chunk_end : nullptr;
    a.split_ws = has_split ? (float *)split->workspace : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const bool low = (double)E < kLowDegree * (double)NS;
    if (dense && dense->bits) {
        // small dense graph: the transposed bit mask on the matrix pipe (node_attn_dense.h), the lean CSR kernel behind it
        // predicated on the range flag
        if (!a.lean || a.masked || low || table_dtype != HAN_DTYPE_F32 || table_gid || edge_val || K != 8 || FP != 8)
            return HAN_E_BADARG;
        DenseArgs d;
        if (!dense_geometry(NS, dense->n_table, false, dense, &d)) return HAN_E_WORKSPACE;
        dense_f2_range_kernel<<<kDenseRangeBlocks, 256, 0, st>>>(f2, NS, d.hdr);
        const dim3 dg((unsigned)((NS + kDenseRowsPerBlock - 1) / kDenseRowsPerBlock), (unsigned)d.S);
        node_attn_bwd_dense_kernel<<<dg, 256, 0, st>>>(a, d);
        node_attn_bwd_dense_finish_kernel<<<(int)((NS + 15) / 16), 256, 0, st>>>(a, d);
        HAN_CHECK_LAUNCH();
        a.only_if = reinterpret_cast<const int *>(d.hdr) + 16;
    }
    const RowBins bins = bins_of(split);
    if (table_dtype == HAN_DTYPE_BF16) {
        HAN_DISPATCH_FP(FP, { launch_bwd_cols<FPC, true>(a, low, has_split, bins, st); })
    } else {
        HAN_DISPATCH_FP(FP, { launch_bwd_cols<FPC, false>(a, low, has_split, bins, st); })
    }
    HAN_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t han_score_param_bwd_workspace(int64_t N, int K, int FP) {
    (void)N; (void)FP;
    return (size_t)kReduceBlocks * (size_t)(128 + 2 * K) * sizeof(float);
}

extern "C" int han_score_param_bwd(const void *H, int table_dtype, const float *df1, const float *df2,
                                   float *da1, float *da2, float *db1, float *db2, void *workspace,
                                   size_t workspace_bytes, int64_t N, int K, int FP, void *stream) {
    if (!H || !df1 || !df2 || !da1 || !da2 || !db1 || !db2 || !workspace || N < 0) return HAN_E_BADARG;
    if (!fp_supported(K, FP) || !dtype_ok(table_dtype, FP)) return HAN_E_UNSUPPORTED;
    if (workspace_bytes < han_score_param_bwd_workspace(N, K, FP)) return HAN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = han_grid_for(N, 16, kReduceBlocks);
    if (table_dtype == HAN_DTYPE_BF16) {
        HAN_DISPATCH_FP(FP, {
            score_param_bwd_kernel<FPC, true><<<grid, 256, 0, st>>>(H, df1, df2, (float *)workspace, N);
        })
    } else {
        HAN_DISPATCH_FP(FP, {
            score_param_bwd_kernel<FPC, false><<<grid, 256, 0, st>>>(H, df1, df2, (float *)workspace, N);
        })
    }
    HAN_CHECK_LAUNCH();
    HanReduceOut o = han_reduce_to(da1, 128 + 2 * K);
    o.ptr[1] = da2; o.ptr[2] = db1; o.ptr[3] = db2;
    o.seg_end[0] = 64; o.seg_end[1] = 128; o.seg_end[2] = 128 + K; o.seg_end[3] = 128 + 2 * K;
    o.nseg = 4;
    hipError_t e = han_reduce_slabs((const float *)workspace, grid, 128 + 2 * K, 128 + 2 * K, o, st);
    if (e != hipSuccess) return (int)e;
    return 0;
}

extern "C" int han_node_attn_coefs(const int64_t *rowptr, const int32_t *colidx, const float *edge_val,
                                   const int32_t *table_gid, const float *f1, const float *f2, float *coef,
                                   int mean_heads, int64_t N, int64_t E, int K, int FP, float slope,
                                   float coef_drop, uint64_t seed, const uint64_t *seed_dev, int64_t row_offset,
                                   void *stream) {
    if (!rowptr || (!colidx && E > 0) || !f1 || !f2 || (!coef && E > 0) || N < 0 || E < 0) return HAN_E_BADARG;
    if (!fp_supported(K, FP)) return HAN_E_UNSUPPORTED;
    if (coef_drop < 0.f || coef_drop >= 1.f) return HAN_E_BADARG;
    if (N == 0 || E == 0) return 0;
    CoefArgs a;
    a.rowptr = rowptr; a.colidx = colidx; a.edge_val = edge_val; a.gid = table_gid; a.f1 = f1; a.f2 = f2;
    a.coef = coef; a.N = N; a.slope = slope; a.mean_heads = mean_heads;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_dev = seed_dev;
    a.thr_coef = coef_drop > 0.f ? han_keep_threshold(1.f - coef_drop) : HAN_KEEP_ALL;
    a.inv_keep_coef = 1.f / (1.f - coef_drop);
    a.row_offset = row_offset;
    const int grid = attn_grid(N);
    hipStream_t st = (hipStream_t)stream;
    HAN_DISPATCH_FP(FP, { node_attn_coef_kernel<HAN_D / FPC><<<grid, 256, 0, st>>>(a); })
    HAN_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t han_node_attn_dense_workspace(int64_t rows, int64_t n_table, int train) {
    if (rows <= 0 || n_table <= 0) return 0;
    return dense_workspace_bytes(rows, n_table, train ? DenseRowWidth<true>::value : DenseRowWidth<false>::value);
}

extern "C" int han_csr_to_bitmask(const int64_t *rowptr, const int32_t *colidx, int64_t N, int64_t n_table,
                                  uint32_t *bits, int64_t ld_words, int *repeated, void *stream) {
    if (!rowptr || !bits || N < 0 || n_table <= 0 || ld_words < (n_table + 31) / 32) return HAN_E_BADARG;
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(bits, 0, (size_t)N * (size_t)ld_words * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    if (repeated) {
        e = hipMemsetAsync(repeated, 0, sizeof(int), st);
        if (e != hipSuccess) return (int)e;
    }
    csr_to_bitmask_kernel<<<attn_grid(N), 256, 0, st>>>(rowptr, colidx, N, n_table, bits, ld_words, repeated);
    HAN_CHECK_LAUNCH();
    return 0;
}
