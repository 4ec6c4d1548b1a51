// K2 on the matrix pipe for SMALL, DENSE meta-path graphs (the reference's own data sets: ACM PSP 24 % of all pairs,
// DBLP APCPA / APTPA 30 % / 78 %, a few thousand nodes) -- included by node_attn.hip inside its anonymous namespace.
//
// The reference computes these heads densely: an N x N logit matrix, an additive -1e9 mask, a softmax over rows and
// matmul(coefs, seq_fts) (utils/layers.py:26-34).  The CSR kernels spend 40-130 M vector instructions per launch on
// such graphs (profiles/r03_pmc_k2_small_dense.json): every stored entry costs an exp, an online-softmax step and
// 8 + 8 FMAs per head in the vector ALU.  Here the products alpha . H run on v_mfma_f32_16x16x4_f32 (exact fp32), and
// the exponential disappears from the inner loop:
//
//     exp(LeakyReLU_0.2(x)) = max(e^x, e^{0.2 x}),   x = f1_i + f2_j
//  => p_ij = exp(LeakyReLU(x) - m_i) = max(A_i B_j, C_i D_j)
//     B_j = e^{f2_j - F},  D_j = e^{0.2 (f2_j - F)}                       per table row and head   (<= 1)
//     A_i = e^{f1_i + F - m_i},  C_i = e^{0.2 (f1_i + F) - m_i}           per destination and head (<= 1)
//     F = max_j f2_j (per head, over the whole table),  m_i = LeakyReLU(f1_i + F) >= every logit of row i
//
// -- a FIXED shift per row instead of a running maximum, so partial sums of different column ranges simply add.
// Every factor is <= 1 (no overflow); p >= e^{-(F - min_j f2_j)}, so the sums keep full relative precision as long as
// the per-head range of f2 stays below 80 (dense_f2_range_kernel checks; beyond it -- softmax rows that are one-hot to
// 35 decimal places -- the caller's lean CSR kernel runs instead, predicated on the flag this path leaves in its header).
//
// MFMA map (16x16x4, A[l&15][l>>4], B[l>>4][l&15], D col = l&15, rows 4 (l>>4) + r): M = 16 destination rows, K = 4
// table rows per step, N = the 8 columns of ONE head (columns 8..15 of the product are not used: A differs per head,
// so a head cannot share an instruction with another one -- 78 TF of useful fp32 products at peak).  A lane computes
// exactly the p it feeds: (i = l & 15, j = 4 s + (l >> 4)); the adjacency is a bit mask (one 32-bit word per row and
// 32-column tile), applied as a sign-extended-bit AND.
//
// Work split: a block of 4 waves owns 64 destination rows x one segment of the column tiles; H rows, the (B, D) pairs
// and nothing else are staged through LDS per 32-column tile; the segments' partial sums go to a slab and a finishing
// launch adds them in segment order (fixed order: bitwise reproducible), normalises and applies bias / activation
// (write_row, as the CSR kernels).  fp32 tables, 8 heads x 8 columns, binary adjacency without repeated entries,
// table index == global id.

constexpr int kDenseTile = 32;          // table rows per staged tile (= bits of one mask word)
constexpr int kDenseHLd = 72;           // floats per staged H row: 64 + 8 (rows j, j + 1 of a 32-lane half land on disjoint banks)
constexpr int kDenseRowsPerBlock = 64;
constexpr float kDenseMaxRange = 80.f;

struct DenseArgs {
    const uint32_t *bits;    // [rows][ldw] adjacency bit mask: bit (j & 31) of word j >> 5 of row i <=> j is a neighbour of i
    int64_t ldw;
    float *slab;             // [S][rows][row_width] partial sums per segment
    float *hdr;              // [0..7] F = max f2 per head, [8..15] min, [16] (int) 1 = range too wide: run the CSR kernel
                             // (published by block (0, 0) of the dense launch); [32 ..) the range launch's partials
    int S, tiles_per_seg, tiles;
    int64_t NT;              // table rows
};

constexpr int kDenseRangeBlocks = 32;
constexpr int kDenseHdrFloats = 32 + 16 * kDenseRangeBlocks;      // [0..16] final (F, min, flag) | [32 ..) per-block partials

// per head: max / min of f2 over the table, in kDenseRangeBlocks slices (one block each; partial results at
// hdr[32 + 16 b]: 8 max | 8 min).  The consumers fold the 32 partials themselves (dense_range below): as ONE block
// this was 8 us alone and 30 us inside a captured epoch, where a single 1024-thread block waits for a free CU behind
// the other meta-paths' kernels (profiles/r04_dblp_like_graph_kernel_stats.csv of the first build).
__global__ __launch_bounds__(256) void dense_f2_range_kernel(const float *f2, int64_t NT, float *hdr) {
    __shared__ float smx[4][8], smn[4][8];
    const int64_t per = (NT + kDenseRangeBlocks - 1) / kDenseRangeBlocks;
    const int64_t r0 = per * blockIdx.x, r1 = (r0 + per < NT) ? r0 + per : NT;
    float mx[8], mn[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) { mx[h] = -3.0e38f; mn[h] = 3.0e38f; }
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const float4_t a = *reinterpret_cast<const float4_t *>(f2 + r * 8);
        const float4_t b = *reinterpret_cast<const float4_t *>(f2 + r * 8 + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            mx[t] = fmaxf(mx[t], a[t]); mn[t] = fminf(mn[t], a[t]);
            mx[4 + t] = fmaxf(mx[4 + t], b[t]); mn[4 + t] = fminf(mn[4 + t], b[t]);
        }
    }
#pragma unroll
    for (int h = 0; h < 8; ++h)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], o, 64));
            mn[h] = fminf(mn[h], __shfl_xor(mn[h], o, 64));
        }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int h = 0; h < 8; ++h) { smx[w][h] = mx[h]; smn[w][h] = mn[h]; }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int h = threadIdx.x;
        float *part = hdr + 32 + 16 * blockIdx.x;
        part[h] = fmaxf(fmaxf(smx[0][h], smx[1][h]), fmaxf(smx[2][h], smx[3][h]));
        part[8 + h] = fminf(fminf(smn[0][h], smn[1][h]), fminf(smn[2][h], smn[3][h]));
    }
}

// Fold the partial ranges (every block of a dense launch does, identically): sF[0..7] = F = max f2 per head; returns
// whether the fixed-shift form applies (every head's range within kDenseMaxRange; NaN / inf scores fail the comparison
// too and go to the CSR kernels, which treat them as the reference does).  Block (0, 0) publishes F and the verdict in
// hdr[0..16] for the launches behind this one (the finishing pass; the predicated CSR kernel).  Contains a barrier.
__device__ __forceinline__ bool dense_range(float *hdr, float *sF) {
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    if (threadIdx.x < 8) {
        const int h = threadIdx.x;
        float M = -3.0e38f, m = 3.0e38f;
#pragma unroll 8
        for (int b = 0; b < kDenseRangeBlocks; ++b) {
            M = fmaxf(M, hdr[32 + 16 * b + h]);
            m = fminf(m, hdr[32 + 16 * b + 8 + h]);
        }
        sF[h] = M;
        const bool ok = (M - m) <= kDenseMaxRange && fabsf(M) < 1.0e30f;
        if (!ok) atomicOr(&s_bad, 1);
        if (blockIdx.x == 0 && blockIdx.y == 0) { hdr[h] = M; hdr[8 + h] = m; }
    }
    __syncthreads();
    const bool bad = s_bad != 0;
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) reinterpret_cast<int *>(hdr)[16] = bad ? 1 : 0;
    return !bad;
}

// (Folding the finishing launch into the last segment to arrive for a row block -- a ticket per block, __threadfence()
// around it -- was built and measured: every one of the ~1000 blocks then pays a device-scope release, i.e. a write-back
// of its XCD's L2, and the eval launch went from 94 to 212 us.  The finishing pass stays a launch of its own.)

template <bool TRAIN>
struct DenseRowWidth { static constexpr int value = TRAIN ? 144 : 72; };      // acc 64 | [accp 64] | l 8 | [tl 8]

// one 16-lane group per destination row: the segments' partial sums in segment order, then write_row
template <bool TRAIN>
__global__ __launch_bounds__(256) void node_attn_fwd_dense_finish_kernel(const FwdArgs a, const DenseArgs d) {
    if (reinterpret_cast<const int *>(d.hdr)[16]) return;
    constexpr int RW = DenseRowWidth<TRAIN>::value;
    const int q = threadIdx.x & 15;
    const int head = q >> 1;
    const float4_t c4 = *reinterpret_cast<const float4_t *>(a.c + 4 * q);
    {
        const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
        if (row >= a.N) return;
        RowState<TRAIN> st;
        st.init();
        for (int sgm = 0; sgm < d.S; ++sgm) {
            const float *r = d.slab + ((int64_t)sgm * a.N + row) * RW;
            const float4_t v = *reinterpret_cast<const float4_t *>(r + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) st.acc[t] += v[t];
            st.l += r[(TRAIN ? 128 : 64) + head];
            if (TRAIN) {
                const float4_t vp = *reinterpret_cast<const float4_t *>(r + 64 + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) st.accp[t] += vp[t];
                st.tl += r[136 + head];
            }
        }
        st.m = han_lrelu(a.f1[row * 8 + head] + d.hdr[head], a.slope);      // the row's fixed shift: lse = m + log l
        write_row<8, TRAIN>(a, row, st, q, head, c4, true);
    }
}

// grid (ceil(N / 64), S); block 256 = 4 waves x one 16-row tile each
template <bool TRAIN>
__global__ __launch_bounds__(256) void node_attn_fwd_dense_kernel(const FwdArgs a_in, const DenseArgs d) {
    __shared__ float sF[8];
    if (!dense_range(d.hdr, sF)) return;      // range too wide (every block agrees): the predicated CSR launch does the work
    FwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int RW = DenseRowWidth<TRAIN>::value;
    __shared__ __attribute__((aligned(16))) float Hs[kDenseTile * kDenseHLd];
    __shared__ __attribute__((aligned(16))) float2 E2s[kDenseTile * 8];
    __shared__ float F2s[TRAIN ? kDenseTile * 8 : 1];      // training: the raw scores, for the sign of x = f1_i + f2_j
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * kDenseRowsPerBlock + 16 * w;
    const int64_t irow = (i0 + n < a.N) ? i0 + n : a.N - 1;       // the destination this lane computes p for
    const bool drop_c = TRAIN && a.thr_coef < HAN_KEEP_ALL;
    const bool drop_f = TRAIN && a.lsb_mask;
    const float *Hf = reinterpret_cast<const float *>(a.H);

    float Ai[8], Ci[8], f1i[8];
    {
        const float4_t fa = *reinterpret_cast<const float4_t *>(a.f1 + irow * 8);
        const float4_t fb = *reinterpret_cast<const float4_t *>(a.f1 + irow * 8 + 4);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            f1i[h] = h < 4 ? fa[h & 3] : fb[h & 3];
            const float x = f1i[h] + sF[h];
            const float m = han_lrelu(x, a.slope);
            Ai[h] = __expf(x - m);
            Ci[h] = __expf(a.slope * x - m);
        }
    }
    float4_t acc[8], accp[8];      // (accp / tls: training only; dead otherwise)
    float ls[8], tls[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        acc[h] = (float4_t){0.f, 0.f, 0.f, 0.f};
        accp[h] = (float4_t){0.f, 0.f, 0.f, 0.f};
        ls[h] = 0.f;
        tls[h] = 0.f;
    }
    const uint32_t gi = (uint32_t)(irow + a.row_offset);
    const int t0 = blockIdx.y * d.tiles_per_seg;
    const int t1 = (t0 + d.tiles_per_seg < d.tiles) ? t0 + d.tiles_per_seg : d.tiles;
    // staging registers of the NEXT tile: 32 table rows x 64 floats (two float4 per thread: row idx >> 4, quad idx & 15),
    // one score per thread (row tid >> 3, head tid & 7) and this lane's mask word -- loaded one tile ahead, so that the L2
    // round trip runs under the previous tile's MFMAs
    float4_t hreg[2];
    float f2reg;
    uint32_t wreg;
    auto load_tile = [&](int jt) {
        const int jc = jt < d.tiles ? jt : d.tiles - 1;      // the last prefetch reads a tile that exists and is not used
        const int64_t j0 = (int64_t)jc * kDenseTile;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + 256 * u;
            const int64_t j = (j0 + (idx >> 4) < d.NT) ? j0 + (idx >> 4) : d.NT - 1;
            hreg[u] = *reinterpret_cast<const float4_t *>(Hf + j * HAN_D + 4 * (idx & 15));
        }
        const int64_t j = (j0 + (tid >> 3) < d.NT) ? j0 + (tid >> 3) : d.NT - 1;
        f2reg = a.f2g[j * 8 + (tid & 7)];
        wreg = d.bits[irow * d.ldw + jc];
    };
    const float fmax_h = sF[tid & 7];
    load_tile(t0);
    for (int jt = t0; jt < t1; ++jt) {
        const int64_t j0 = (int64_t)jt * kDenseTile;
        __syncthreads();      // the previous tile's reads are done
        {   // stage the prefetched tile; rows are masked by their keep bits in training (layers.py:31-32)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = tid + 256 * u;
                float4_t v = hreg[u];
                if (drop_f) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int bits = __float_as_int(v[t]);
                        v[t] = __int_as_float(bits & han_bit_mask<0>(bits));
                    }
                }
                *reinterpret_cast<float4_t *>(Hs + (idx >> 4) * kDenseHLd + 4 * (idx & 15)) = v;
            }
            const float y = f2reg - fmax_h;
            E2s[tid] = make_float2(__expf(y), __expf(a.slope * y));      // E2s[(tid >> 3) * 8 + (tid & 7)]
            if constexpr (TRAIN) F2s[tid] = f2reg;
        }
        const uint32_t word = wreg;
        __syncthreads();
        load_tile(jt + 1);      // in flight under this tile's MFMAs
        // One step = the 4 table rows 4 s + q against the 8 heads: 16 (24) LDS reads, ~6 (12) vector instructions and one
        // (two) MFMAs per head.  A ROLLED loop: fully unrolled, the scheduler hoists all 128 LDS reads of a tile to the top
        // (190 + 50 registers).  The MFMA destination is the VGPR form (_lib.EXTRA_FLAGS): in the AGPR form hipcc rotated
        // the 8-16 accumulators of this loop through v_accvgpr_read / _mov / _write every iteration.  Measured at the DBLP
        // APTPA shape (profiles/r04_k2_dense_experiments.md): 62-64 us for the eval launch = 42 us with the per-pair vector
        // work removed (matrix pipe + LDS + barriers; 31 us of pure MFMA issue) + 22 us of vector issue that does NOT hide
        // under the MFMAs -- neither interleaving the heads' vector work with their MFMAs (sched_group_barrier), nor reading
        // a step's LDS values one step ahead, nor 1 / 3 / 4 waves per SIMD moved it (81 / 64 / 62 us).
#pragma unroll 1
        for (int s = 0; s < 8; ++s) {
            const int jr = 4 * s + q;
            const int mbit = __builtin_amdgcn_sbfe((int)word, jr, 1);      // bit jr of the word, sign-extended: 0 or -1
            uint32_t hx[2] = {0u, 0u}, hy[2] = {0u, 0u};
            if (drop_c) {
                const uint32_t gj = (uint32_t)(j0 + jr);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF, gi, gj * 2u + (uint32_t)c);
                    hx[c] = rn.x;
                    hy[c] = rn.y;
                }
            }
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const float2 e = E2s[jr * 8 + h];
                const float ab = Ai[h] * e.x, cd = Ci[h] * e.y;
                float p = fmaxf(ab, cd);
                p = __int_as_float(__float_as_int(p) & mbit);
                ls[h] += p;
                const float b = Hs[jr * kDenseHLd + 8 * h + (n & 7)];
                if constexpr (!TRAIN) {
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, b, acc[h], 0, 0, 0);
                } else {
                    // LeakyReLU'(x) from x itself, as every other kernel of the path takes it (the products ab / cd
                    // would decide x ~ 0 by their rounding)
                    const float psg = (f1i[h] + F2s[jr * 8 + h]) > 0.f ? p : a.slope * p;
                    tls[h] += psg;
                    float pd = p, pds = psg;
                    if (drop_c) {
                        const uint32_t wsel = (h & 2) ? hy[h >> 2] : hx[h >> 2];
                        const uint32_t f = (h & 1) ? (wsel >> 16) : (wsel & 0xFFFFu);
                        const bool keep = f < a.thr_coef;
                        pd = keep ? p : 0.f;
                        pds = keep ? psg : 0.f;
                    }
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(pd, b, acc[h], 0, 0, 0);
                    accp[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(pds, b, accp[h], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // partial sums of this segment.  acc: lane (n, q) holds rows 4 q + r of column 8 h + n (n < 8);
    // ls / tls: lane (row n, q) holds the sum over its own columns -> add the four q groups (fixed order)
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        ls[h] += __shfl_xor(ls[h], 16, 64);
        ls[h] += __shfl_xor(ls[h], 32, 64);
        if (TRAIN) {
            tls[h] += __shfl_xor(tls[h], 16, 64);
            tls[h] += __shfl_xor(tls[h], 32, 64);
        }
    }
    float *seg = d.slab + (int64_t)blockIdx.y * a.N * RW;
    if (n < 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = i0 + 4 * q + r;
            if (i < a.N) {
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    seg[i * RW + 8 * h + n] = acc[h][r];
                    if (TRAIN) seg[i * RW + 64 + 8 * h + n] = accp[h][r];
                }
            }
        }
    }
    if (q == 0 && i0 + n < a.N) {
        float *row = seg + (i0 + n) * RW + (TRAIN ? 128 : 64);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            row[h] = ls[h];
            if (TRAIN) row[8 + h] = tls[h];
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Backward over the TRANSPOSED bit mask (rows = sources j, bits = destinations i): the dense form of
// node_attn_bwd_cols.  Per head:  acc_j += alpha~_ij g_i  (16 sources x 4 destinations x 8 columns per MFMA, A = alpha~),
// and the score gradient  df2_j += alpha_ij LeakyReLU'(x_ij) (am_ij g_i . H~_j - s_i)  needs the dot products
// g_i . H~_j of every pair -- a second small GEMM, G_h (16 i x 8) . H~_h^T (8 x 16 j), two MFMAs per head and 16 x 16 tile
// whose result lands exactly where the lane that owns (i, j) needs it: lane (j = l & 15, q = l >> 4) receives the rows
// i = 4 q + r, so step r of the main product takes destination 4 q + r of the tile (the order of a sum is free).
// alpha_ij = max(A'_i B_j, C'_i D_j) with A'_i = e^{f1_i + F - lse_i}, C'_i = e^{0.2 (f1_i + F) - lse_i} (computed when a
// destination tile is staged) and the source's B_j, D_j in registers.  Same segment / slab / finishing structure as the
// forward; write_src finishes a row.
// ---------------------------------------------------------------------------------------------
constexpr int kDenseGLd = 66;           // floats per staged g row: banks 2 r + {0, 1} for the dot product's A operand
                                        // (16 rows x 2 columns per 32-lane half), rows 4 apart 8 banks apart for B
constexpr int kDenseBwdRowWidth = 72;   // acc 64 | df2 8

__global__ __launch_bounds__(256) void node_attn_bwd_dense_finish_kernel(const BwdColsArgs a, const DenseArgs d) {
    if (reinterpret_cast<const int *>(d.hdr)[16]) return;
    const int q = threadIdx.x & 15;
    const int head = q >> 1;
    const float4_t a14 = *reinterpret_cast<const float4_t *>(a.a1 + 4 * q);
    const float4_t a24 = *reinterpret_cast<const float4_t *>(a.a2 + 4 * q);
    {
        const int64_t src = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
        if (src >= a.NS) return;
        const SrcRow sr = load_src<8, false>(a, src, q, head);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float dfacc = 0.f;
        for (int sgm = 0; sgm < d.S; ++sgm) {
            const float *r = d.slab + ((int64_t)sgm * a.NS + src) * kDenseBwdRowWidth;
            const float4_t v = *reinterpret_cast<const float4_t *>(r + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += v[t];
            dfacc += r[64 + head];
        }
        write_src<8>(a, src, sr, acc, dfacc, q, head, a14, a24);
    }
}

__global__ __launch_bounds__(256) void node_attn_bwd_dense_kernel(const BwdColsArgs a_in, const DenseArgs d) {
    __shared__ float sF[8];
    if (!dense_range(d.hdr, sF)) return;
    BwdColsArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int RB = GsRow<8, false>::bytes;
    constexpr int GB = GsRow<8, false>::g_bytes;
    __shared__ __attribute__((aligned(16))) float Gs[kDenseTile * kDenseGLd];
    __shared__ __attribute__((aligned(16))) float4_t St[kDenseTile * 9];      // (A', C', s, f1) per destination and head; rows of
                                                                              // 9 quads: destinations 4 apart on different slots
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int64_t j0 = (int64_t)blockIdx.x * kDenseRowsPerBlock + 16 * w;
    const int64_t jrow = (j0 + n < a.NS) ? j0 + n : a.NS - 1;       // the source this lane computes alpha for
    const bool drop_c = a.thr_coef < HAN_KEEP_ALL;
    const float *Hf = reinterpret_cast<const float *>(a.H);
    const char *gsb = reinterpret_cast<const char *>(a.gs);

    float Bj[8], Dj[8], f2j[8], Hq[8][2];
    {
        const float4_t fa = *reinterpret_cast<const float4_t *>(a.f2 + jrow * 8);
        const float4_t fb = *reinterpret_cast<const float4_t *>(a.f2 + jrow * 8 + 4);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            f2j[h] = h < 4 ? fa[h & 3] : fb[h & 3];
            const float y = f2j[h] - sF[h];
            Bj[h] = __expf(y);
            Dj[h] = __expf(a.slope * y);
#pragma unroll
            for (int t = 0; t < 2; ++t) {      // H~_j[8 h + 4 t + q]: the B operand of the dot product
                float v = Hf[jrow * HAN_D + 8 * h + 4 * t + q];
                if (a.lsb_mask) v *= han_keep_bit<false>(v) ? a.inv_keep_fts : 0.f;
                Hq[h][t] = v;
            }
        }
    }
    const uint32_t gj = (uint32_t)(jrow + a.src_offset);
    float4_t acc[8];
    float dfa[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) { acc[h] = (float4_t){0.f, 0.f, 0.f, 0.f}; dfa[h] = 0.f; }

    const int t0 = blockIdx.y * d.tiles_per_seg;
    const int t1 = (t0 + d.tiles_per_seg < d.tiles) ? t0 + d.tiles_per_seg : d.tiles;
    float4_t greg[2], sreg;
    uint32_t wreg;
    auto load_tile = [&](int it) {
        const int ic = it < d.tiles ? it : d.tiles - 1;
        const int64_t i0 = (int64_t)ic * kDenseTile;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + 256 * u;
            const int64_t i = (i0 + (idx >> 4) < d.NT) ? i0 + (idx >> 4) : d.NT - 1;
            greg[u] = *reinterpret_cast<const float4_t *>(gsb + i * RB + 16 * (idx & 15));
        }
        const int64_t i = (i0 + (tid >> 3) < d.NT) ? i0 + (tid >> 3) : d.NT - 1;
        sreg = *reinterpret_cast<const float4_t *>(gsb + i * RB + GB + 16 * (tid & 7));
        wreg = d.bits[jrow * d.ldw + ic];
    };
    const float fmax_h = sF[tid & 7];
    load_tile(t0);
    for (int it = t0; it < t1; ++it) {
        const int64_t i0 = (int64_t)it * kDenseTile;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {      // 264-B rows: two 8-byte stores per quad
            const int idx = tid + 256 * u;
            float *dst = Gs + (idx >> 4) * kDenseGLd + 4 * (idx & 15);
            *reinterpret_cast<float2 *>(dst) = make_float2(greg[u][0], greg[u][1]);
            *reinterpret_cast<float2 *>(dst + 2) = make_float2(greg[u][2], greg[u][3]);
        }
        {
            const float x = sreg[0] + fmax_h;      // f1_i + F
            float4_t sv;
            sv[0] = __expf(x - sreg[1]);
            sv[1] = __expf(a.slope * x - sreg[1]);
            sv[2] = sreg[2];
            sv[3] = sreg[0];
            St[(tid >> 3) * 9 + (tid & 7)] = sv;
        }
        const uint32_t word = wreg;
        __syncthreads();
        load_tile(it + 1);
#pragma unroll 1
        for (int sub = 0; sub < 2; ++sub) {
            // the four steps' destinations of this lane: i = 16 sub + 4 q + s; mask bits and dropout draws first (shared by
            // the heads), then head by head: dot(i, j) of the 16 x 16 tile (two MFMAs), then the four steps
            int mbit[4];
            uint32_t hx[4][2], hy[4][2];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int il = 16 * sub + 4 * q + s;
                mbit[s] = __builtin_amdgcn_sbfe((int)word, il, 1);
                hx[s][0] = hx[s][1] = hy[s][0] = hy[s][1] = 0u;
                if (drop_c) {
                    const uint32_t gi = (uint32_t)(i0 + il + a.dst_offset);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_COEF, gi, gj * 2u + (uint32_t)c);
                        hx[s][c] = rn.x;
                        hy[s][c] = rn.y;
                    }
                }
            }
            auto dot_of = [&](const int h) {      // g_i . H~_j of head h for this lane's four destinations
                const float g0 = Gs[(16 * sub + n) * kDenseGLd + 8 * h + q];
                const float g1 = Gs[(16 * sub + n) * kDenseGLd + 8 * h + 4 + q];
                float4_t z = {0.f, 0.f, 0.f, 0.f};
                z = __builtin_amdgcn_mfma_f32_16x16x4f32(g0, Hq[h][0], z, 0, 0, 0);
                return __builtin_amdgcn_mfma_f32_16x16x4f32(g1, Hq[h][1], z, 0, 0, 0);
            };
            float4_t dot_next = dot_of(0);
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const float4_t dot = dot_next;
                if (h < 7) dot_next = dot_of(h + 1);      // one head ahead: its two MFMAs run under this head's vector work
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int il = 16 * sub + 4 * q + s;
                    const float4_t sv = St[il * 9 + h];
                    float al = fmaxf(sv[0] * Bj[h], sv[1] * Dj[h]);
                    al = __int_as_float(__float_as_int(al) & mbit[s]);
                    float wv = al;
                    if (drop_c) {
                        const uint32_t wsel = (h & 2) ? hy[s][h >> 2] : hx[s][h >> 2];
                        const uint32_t f = (h & 1) ? (wsel >> 16) : (wsel & 0xFFFFu);
                        wv = f < a.thr_coef ? al * a.inv_keep_coef : 0.f;
                    }
                    const float t = wv * dot[s] - al * sv[2];
                    dfa[h] += (sv[3] + f2j[h]) > 0.f ? t : a.slope * t;
                    const float b = Gs[il * kDenseGLd + 8 * h + (n & 7)];
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, b, acc[h], 0, 0, 0);
                }
                // keep the heads apart: unfenced, the scheduler hoists the LDS reads of all 32 (head, step) pairs to the
                // top (212 + 80 registers, one wave per SIMD)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        dfa[h] += __shfl_xor(dfa[h], 16, 64);
        dfa[h] += __shfl_xor(dfa[h], 32, 64);
    }
    float *seg = d.slab + (int64_t)blockIdx.y * a.NS * kDenseBwdRowWidth;
    if (n < 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t j = j0 + 4 * q + r;
            if (j < a.NS) {
#pragma unroll
                for (int h = 0; h < 8; ++h) seg[j * kDenseBwdRowWidth + 8 * h + n] = acc[h][r];
            }
        }
    }
    if (q == 0 && j0 + n < a.NS) {
#pragma unroll
        for (int h = 0; h < 8; ++h) seg[(j0 + n) * kDenseBwdRowWidth + 64 + h] = dfa[h];
    }
}

