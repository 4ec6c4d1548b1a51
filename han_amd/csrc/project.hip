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
    const unsigned char *wimg;   // bf16 x 6 kernels: the pre-split, LDS-ready image of W (project_wimage_kernel), in the workspace
    const void *X;      // fp32 or bf16 (x_bf16), row stride ldx ELEMENTS
    int64_t ldx;
    const float *W;
    void *H;            // fp32 or bf16 (h_bf16), 64 elements per row
    int x_bf16, h_bf16;
    int64_t N;
    int F;
    uint32_t seed_lo, seed_hi, thr_in, thr_fts;   // thr_fts < 2^16: stamp keep bits into H
    uint32_t fts_stream;                          // HAN_STREAM_FTS + 4 * slice (HAN_FLAG_FTS_SLICE: slices of a wide head)
    const uint64_t *seed_dev;
    float inv_keep_in;
    int64_t row_offset;
    // split-F for short inputs (few row blocks, long reduction): blockIdx.y owns the
    // features [y*f_chunk, (y+1)*f_chunk) and writes a raw partial tile to `partial`
    // ([nsplit][N][64] fp32); project_finish_kernel sums them in a fixed order.
    int f_chunk;        // multiple of BK; >= F when not split
    float *partial;     // null when not split
    // attention scores in the epilogue (layers.py:23-24), from the values exactly as stored; null f1 = not fused
    // (split-F: project_scores_kernel runs after the finish kernel)
    const float *a1, *a2, *b1, *b2;
    float *f1, *f2;
    // keep table of the per-head input dropout for the backward (han_hip.h: han_project_keep_bytes), or null
    uint8_t *keep;
};

// f1 = H_k . a1_k + b1_k, f2 likewise, for the row whose stored elements (row, 16t + l15), t < 4, this lane
// holds in vst[]: the F' columns of a head sit in min(F',16) consecutive lanes of a 16-lane group and in
// max(1, F'/16) column tiles.  Epilogue code (once per output element), not a hot loop.
template <int FP>
__device__ __forceinline__ void tile_scores(const ProjFwdArgs &a, int64_t row, bool row_ok, const float (&vst)[4],
                                            const float (&a1c)[4], const float (&a2c)[4], int l15) {
    constexpr int K = HAN_D / FP;
    constexpr int NT = FP >= 16 ? FP / 16 : 1;       // column tiles per head
    constexpr int NL = FP >= 16 ? 16 : FP;           // lanes per head inside a tile
    float r1[4 / NT], r2[4 / NT];
#pragma unroll
    for (int u = 0; u < 4 / NT; ++u) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = u * NT; t < (u + 1) * NT; ++t) {
            s1 += vst[t] * a1c[t];
            s2 += vst[t] * a2c[t];
        }
        // butterfly over the head's NL lanes on DPP operands (bitwise the __shfl_xor butterfly, without its 3-4
        // ds_bpermute round trips per sum: the fused eval kernel issued 1024 of them per wave)
        if (NL > 1) { s1 += han_dpp_xor16<1>(s1); s2 += han_dpp_xor16<1>(s2); }
        if (NL > 2) { s1 += han_dpp_xor16<2>(s1); s2 += han_dpp_xor16<2>(s2); }
        if (NL > 4) { s1 += han_dpp_xor16<4>(s1); s2 += han_dpp_xor16<4>(s2); }
        if (NL > 8) { s1 += han_dpp_xor16<8>(s1); s2 += han_dpp_xor16<8>(s2); }
        r1[u] = s1;
        r2[u] = s2;
    }
    if (FP == 8) {
        // lanes 0 / 8 of the group hold heads 2u / 2u+1: bring the odd heads over and let lane 0 write the
        // row's 8 scores as two 16-byte stores (16 scattered 4-byte stores per row cost as much as the
        // separate scores kernel they replaced)
        float4_t o1[2], o2[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float e1 = han_dpp_xor16<8>(r1[u]), e2 = han_dpp_xor16<8>(r2[u]);
            o1[u >> 1][2 * (u & 1)] = r1[u] + a.b1[2 * u];
            o1[u >> 1][2 * (u & 1) + 1] = e1 + a.b1[2 * u + 1];
            o2[u >> 1][2 * (u & 1)] = r2[u] + a.b2[2 * u];
            o2[u >> 1][2 * (u & 1) + 1] = e2 + a.b2[2 * u + 1];
        }
        if (row_ok && l15 == 0) {
            float4_t *p1 = reinterpret_cast<float4_t *>(a.f1 + row * K), *p2 = reinterpret_cast<float4_t *>(a.f2 + row * K);
            p1[0] = o1[0]; p1[1] = o1[1];
            p2[0] = o2[0]; p2[1] = o2[1];
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 4 / NT; ++u) {
        const int head = FP >= 16 ? u : (16 * u + l15) / FP;
        if (row_ok && (l15 % NL) == 0) {
            a.f1[row * K + head] = r1[u] + a.b1[head];
            a.f2[row * K + head] = r2[u] + a.b2[head];
        }
    }
}

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

// Four consecutive elements starting at element `idx`, of which the first `nvalid` (1..4) exist: one 16-byte (fp32) /
// 8-byte (bf16) load that needs no more than element alignment when all four exist (global memory is in unaligned
// access mode; F = 1870 of the ACM data set makes every other row start 8 bytes off a 16-byte boundary), element
// loads for a row's last, partial quad.  (Round 3: inputs whose width is not a multiple of 4 used to fall back to
// one 4-byte load per element in the fp32 kernels.)
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int uint2_u __attribute__((ext_vector_type(2), aligned(2)));
__device__ __forceinline__ float4_t load_x4_tail(const void *X, int bf, int64_t idx, int nvalid) {
    float4_t v = {0.f, 0.f, 0.f, 0.f};
    if (nvalid >= 4) {
        if (bf) {
            const uint2_u w = *reinterpret_cast<const uint2_u *>(reinterpret_cast<const uint16_t *>(X) + idx);
            v[0] = __uint_as_float(w[0] << 16);
            v[1] = __uint_as_float(w[0] & 0xFFFF0000u);
            v[2] = __uint_as_float(w[1] << 16);
            v[3] = __uint_as_float(w[1] & 0xFFFF0000u);
        } else {
            const float4_u w = *reinterpret_cast<const float4_u *>(reinterpret_cast<const float *>(X) + idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = w[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (e < nvalid) v[e] = load_x1(X, bf, idx + e);
    }
    return v;
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
                if (row < a.N && k0 + c4 < k_end) v = load_x4_tail(a.X, a.x_bf16, row * a.ldx + k0 + c4, k_end - (k0 + c4));
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
    float a1c[4], a2c[4];
    if (a.f1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a1c[t] = a.a1[16 * t + l15];
            a2c[t] = a.a2[16 * t + l15];
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 16 * (w * MT + m) + l4 * 4 + r;
            float vst[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = acc[m][t][0][r];
#pragma unroll
                for (int hh = 1; hh < HPT; ++hh) v = (myhh == hh) ? acc[m][t][hh][r] : v;
                if (DROP) v *= a.inv_keep_in;
                vst[t] = v;
                if (a.partial) {   // split-F: raw partial sums, finished by project_finish_kernel
                    if (row < a.N) a.partial[((int64_t)blockIdx.y * a.N + row) * HAN_D + 16 * t + l15] = v;
                    continue;
                }
                uint32_t keepbit = 0, stamp = 0;
                if (a.thr_fts < HAN_KEEP_ALL) {
                    // projected-row dropout (layers.py:31-32): the keep bit rides in the lowest
                    // mantissa bit of the STORED element (fp32 bit 0 / bf16 bit 0)
                    const int d = 16 * t + l15;
                    const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, a.fts_stream,
                                                    (uint32_t)(row + a.row_offset), (uint32_t)(d >> 2));
                    keepbit = rn.field(d & 3) < a.thr_fts ? 1u : 0u;
                    stamp = 1;
                }
                if (a.h_bf16) {
                    uint32_t b = han_f32_to_bf16_bits(v);
                    if (stamp) b = (b & ~1u) | keepbit;
                    if (row < a.N) reinterpret_cast<uint16_t *>(a.H)[row * HAN_D + 16 * t + l15] = (uint16_t)b;
                    vst[t] = __uint_as_float(b << 16);
                } else {
                    if (stamp) v = __uint_as_float((__float_as_uint(v) & ~1u) | keepbit);
                    if (row < a.N) reinterpret_cast<float *>(a.H)[row * HAN_D + 16 * t + l15] = v;
                    vst[t] = v;
                }
            }
            if (a.f1) tile_scores<FP>(a, row, row < a.N, vst, a1c, a2c, l15);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Forward on the bf16 matrix pipe with fp32-class accuracy ("bf16 x 6").
//
// An fp32 number splits EXACTLY into three bf16 terms by truncation, x = hi + mid + lo (8 + 8 + 8
// significand bits; every subtraction below is exact).  With both operands split, the six products
//     hi*hi' + hi*mid' + mid*hi' + hi*lo' + lo*hi' + mid*mid'
// (each bf16 x bf16 product is exact in the fp32 accumulator) leave out only mid*lo', lo*mid', lo*lo':
// < 2^-23 of |x w| per term, the size of an fp32 rounding -- against 6 MFMAs of
// v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32) where the exact-fp32 pipe needs 8
// v_mfma_f32_16x16x4_f32 of 32 cycles: 2.7x less matrix time, and the per-head input-dropout masks
// (layers.py:18-19; two heads per 16-column tile at F' = 8, so every tile is issued twice) become
// packed 16-bit ANDs on the A fragments instead of fp32 selects.
// Block: 128 rows x all 64 columns, K-step 32; X / W tiles are split while they are staged into LDS
// (X row-major [row][k], W transposed [col][k], 96-B rows) so that a lane's 8 k-values are one
// 16-B read.  DROP is built for F' = 8 (the reference shape); without dropout any head shape runs.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

constexpr int B6_ROWS = 128;             // rows per block
constexpr int B6_LDB = 64;               // bytes per LDS row: 32 bf16, NO padding, 16-byte slots XOR-swizzled (b6_swz)
constexpr int B6_XBYTES = B6_ROWS * B6_LDB;
constexpr int B6_WBYTES = HAN_D * B6_LDB;
constexpr int B6_WTILE = 3 * B6_WBYTES;  // bytes of one K-step of the pre-split W image (3 terms x 64 columns x 64 B)

// LDS layout of the X / W tiles (round 4; round 3 used 96-B padded rows, conflict-free for the fragment READS but
// 4-way conflicting on the W staging stores and 2-way on the X stores -- ds_write banks are (a/4) % 32 --, and the
// keep-word collection was 16-way: profiles/r03_pmc_k1.json, 43 % of the LDS cycles were conflict cycles).
// A row is 64 B = four 16-byte slots; slot sl of row r sits at slot sl ^ b6_swz(r).  ds_read_b128 is served in the
// lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63} (MI355X_MICROARCH.md,
// LDS) over 64 banks = four rows: a fragment read (lane: row l15, slot l4) conflicts only among rows equal mod 4,
// and within a group those rows carry l4 in {0,1,1,0} / {1,0,0,1} / {2,3,3,2} / {3,2,2,3} for (r >> 2) = 0..3 -- the
// table 0,2,3,1 makes the four slots distinct in every group.  The X staging store (8 B per lane, 16 contiguous
// lanes = two whole rows) then covers 128 contiguous bytes: conflict-free too.
__device__ __forceinline__ int b6_swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
// byte offset of 16-byte slot `sl` of row `row` inside a tile
__device__ __forceinline__ int b6_off(int row, int sl) { return row * B6_LDB + ((sl ^ b6_swz(row)) << 4); }

// x == h + m + l exactly; each term has <= 8 significand bits (its low 16 bits are zero)
__device__ __forceinline__ void b6_split(float x, uint32_t &h, uint32_t &m, uint32_t &l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));
}
// two truncated terms -> one packed bf16 pair (element 0 in the low half)
__device__ __forceinline__ uint32_t b6_pack(uint32_t e0, uint32_t e1) {
    return __builtin_amdgcn_perm(e1, e0, 0x07060302u);
}
__device__ __forceinline__ f32x4 b6_mfma(const i32x4 &a, const i32x4 &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
}
// per 16-bit field f of w: 0xFFFF if f < thr else 0 (thr < 2^16 in both halves of `thr2`):
// saturating thr - f is non-zero exactly when f < thr; min(.,1) -> 0/1; 0 - that -> 0 / 0xFFFF
// (three packed 16-bit instructions for two fields)
__device__ __forceinline__ uint32_t b6_keep_masks(uint32_t w, uint32_t thr2, uint32_t one2) {
    // inline asm: hipcc turns the vector-typed form into per-field v_cmp + v_cndmask + re-pack
    uint32_t s, m;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(s) : "v"(thr2), "v"(w));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(s), "v"(one2));
    asm("v_pk_sub_u16 %0, 0, %1" : "=v"(s) : "v"(m));
    return s;
}

// W (P x F x 64 fp32) -> the LDS-ready image the bf16 x 6 kernels copy per K-step: [K-step kt][meta-path P][term 3][column 64][64 B],
// the 32 k-values of a column as bf16 in four 16-byte slots, swizzled as in LDS (b6_off); k >= F is zero.  Every block
// of the forward used to split the same W tiles again (7813 blocks x 8 K-steps at SYN-1M) and store them transposed
// with 4-way bank conflicts; now a block copies 8-byte pieces straight through (16 contiguous lanes = 128 contiguous
// bytes: conflict-free).  One thread per (K-step, column, 4 consecutive k).
__global__ __launch_bounds__(256) void project_wimage_kernel(const float *W, unsigned char *img, int F, int ktiles, int P) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_p = (int64_t)ktiles * 64 * 8;
    if (t >= per_p * P) return;
    const int p = (int)(t / per_p);
    const int r = (int)(t % per_p);
    const int kt = r / 512, col = r & 63, kq = (r >> 6) & 7;      // 64 consecutive threads: 64 consecutive columns of one k row
    const float *Wp = W + (int64_t)p * F * HAN_D;
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = kt * 32 + kq * 4 + j;
        b6_split(k < F ? Wp[(int64_t)k * HAN_D + col] : 0.f, h[j], m[j], l[j]);
    }
    unsigned char *dst = img + ((int64_t)kt * P + p) * B6_WTILE + b6_off(col, kq >> 1) + (kq & 1) * 8;
    *reinterpret_cast<uint2 *>(dst) = make_uint2(b6_pack(h[0], h[1]), b6_pack(h[2], h[3]));
    *reinterpret_cast<uint2 *>(dst + B6_WBYTES) = make_uint2(b6_pack(m[0], m[1]), b6_pack(m[2], m[3]));
    *reinterpret_cast<uint2 *>(dst + 2 * B6_WBYTES) = make_uint2(b6_pack(l[0], l[1]), b6_pack(l[2], l[3]));
}

// NW = waves per block (the block always owns 128 rows): 8 waves of one 16-row tile each keep the kernel
// within 128 registers, i.e. FOUR waves per SIMD -- measured round 3: the 4-wave form (248 registers, two
// waves per SIMD) spends 55 % of its wave cycles waiting (profiles/r02_pmc_k1_b6.json: each wave alone
// issues a vector instruction every ~4.8 cycles at best, tools/micro/ubench.hip), and its KEEP variant at
// 260 registers / ONE wave per SIMD ran 1.17 ms against 0.80 ms.
template <bool DROP, bool XBF, bool KEEP, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void project_fwd_b6_kernel(const ProjFwdArgs a_in) {
    ProjFwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int NX = XBF ? 1 : 3;       // split terms of X (a bf16 X is its own high term)
    constexpr int HPT = DROP ? 2 : 1;     // heads per 16-column tile (F' = 8 when DROP)
    constexpr int MT = 8 / NW;            // 16-row tiles per wave
    constexpr int NT = 64 * NW;           // threads
    constexpr int XV = 1024 / NT;         // float4 loads of X per thread per tile (128 rows x 32 k)
    constexpr int WU = (B6_WTILE / 8) / NT;   // 8-byte pieces of the W image per thread per K-step
    __shared__ __attribute__((aligned(16))) unsigned char lds[NX * B6_XBYTES + 3 * B6_WBYTES + (KEEP ? B6_ROWS * 128 : 0)];
    unsigned char *Xs = lds;                          // [NX][128][64 B, slots swizzled]
    unsigned char *Ws = lds + NX * B6_XBYTES;         // [3][64][64 B]   (transposed: [col][k]; a copy of the image's K-step)
    // KEEP: the keep words of four K-steps (128 features = one 128-B line per row) are collected here and leave
    // as whole lines; 8-byte stores per lane and K-step (32-B pieces, queued in front of the next tile's loads)
    // cost the training forward 0.12 ms
    unsigned char *Kacc = lds + NX * B6_XBYTES + 3 * B6_WBYTES;      // [128 rows][128 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * B6_ROWS;
    auto flush_keep = [&](int kbase) {                // rows x features [kbase, kbase + 128) of the table
#pragma unroll
        for (int i = 0; i < 1024 / NT; ++i) {
            const int c = tid + NT * i;
            const int r = c >> 3, sg = c & 7;
            const int64_t krow = row0 + r;
            if (krow < a.N && kbase + 16 * sg < a.F) {      // F % 8 == 0: a 16-B piece may end 8 B past the row
                // the 8-byte pieces of a row are XOR-swizzled by the row (conflict-free collection, see below): the pair
                // (2 sg, 2 sg + 1) sits in the 16-byte slot sg ^ (r >> 1 & 7), its halves exchanged when r is odd
                const uint4 raw = *reinterpret_cast<const uint4 *>(Kacc + r * 128 + 16 * (sg ^ ((r >> 1) & 7)));
                const uint4 v = (r & 1) ? make_uint4(raw.z, raw.w, raw.x, raw.y) : raw;
                uint8_t *dst = a.keep + krow * (int64_t)a.F + kbase + 16 * sg;
                if (kbase + 16 * sg + 8 < a.F) *reinterpret_cast<uint4 *>(dst) = v;
                else *reinterpret_cast<uint2 *>(dst) = make_uint2(v.x, v.y);
            }
        }
    };

    f32x4 acc[MT][4][HPT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int hh = 0; hh < HPT; ++hh) acc[m][t][hh] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging registers: X XV x float4 (row = idx >> 3, k4 = (idx & 7) * 4), W WU 8-byte pieces of the image
    float4_t xr[XV];
    uint2 wq[WU];
    const int ktiles = (a.F + 31) >> 5;
    auto load_tile = [&](int k0) {
        // unconditional loads from clamped addresses + selects: a predicated load becomes a branch
        // around it with its own s_waitcnt, which serialises the tile's loads
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int idx = tid + NT * i;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            const int64_t row = row0 + r;
            const int64_t rc = row < a.N ? row : a.N - 1;
            const int kc = k0 + c4 < a.F ? k0 + c4 : a.F - 4;       // F % 4 == 0 (vec path)
            xr[i] = load_x4(a.X, XBF, rc * a.ldx + kc);             // RAW: out-of-range elements are zeroed when the
        }                                                           // tile is staged -- a select here makes the wave
        const int kt = (k0 >> 5) < ktiles ? (k0 >> 5) : ktiles - 1; // wait for the load BEFORE its MFMA phase (round 3:
        const unsigned char *src = a.wimg + (int64_t)kt * B6_WTILE; // that was the case, every K-step paid the HBM latency)
#pragma unroll
        for (int i = 0; i < WU; ++i) wq[i] = *reinterpret_cast<const uint2 *>(src + 8 * (tid + NT * i));
    };
    const uint32_t thr2 = (a.thr_in & 0xFFFFu) * 0x00010001u, one2 = 0x00010001u;   // keep iff field < thr_in
    load_tile(0);
    for (int k0 = 0; k0 < a.F; k0 += 32) {
        __syncthreads();   // the previous tile's fragment reads are done
        if constexpr (KEEP) {
            if (k0 > 0 && (k0 & 96) == 0) flush_keep(k0 - 128);      // the four K-steps before this one are complete
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int idx = tid + NT * i;
            const int off = b6_off(idx >> 3, (idx & 7) >> 1) + (idx & 1) * 8;
            const bool ok = row0 + (idx >> 3) < a.N && k0 + (idx & 7) * 4 < a.F;
            uint32_t h[4], m[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) b6_split(ok ? xr[i][e] : 0.f, h[e], m[e], l[e]);
            *reinterpret_cast<uint2 *>(Xs + off) = make_uint2(b6_pack(h[0], h[1]), b6_pack(h[2], h[3]));
            if (!XBF) {
                *reinterpret_cast<uint2 *>(Xs + B6_XBYTES + off) = make_uint2(b6_pack(m[0], m[1]), b6_pack(m[2], m[3]));
                *reinterpret_cast<uint2 *>(Xs + 2 * B6_XBYTES + off) = make_uint2(b6_pack(l[0], l[1]), b6_pack(l[2], l[3]));
            }
        }
#pragma unroll
        for (int i = 0; i < WU; ++i) *reinterpret_cast<uint2 *>(Ws + 8 * (tid + NT * i)) = wq[i];      // straight copy
        __syncthreads();
        // UNCONDITIONAL (clamped addresses): under an `if` the loaded registers meet the not-loaded path in a phi, whose
        // copies make the wave wait for the loads right here instead of after the MFMA phase
        load_tile(k0 + 32);   // in flight under the MFMAs
        const int fro = l15 * B6_LDB + ((l4 ^ b6_swz(l15)) << 4);      // this lane's fragment inside a 16-row tile
        auto bfrag = [&](int t, int s3) {        // B[k = 8*l4 + j][col = 16t + l15]
            return *reinterpret_cast<const i32x4 *>(Ws + s3 * B6_WBYTES + 16 * t * B6_LDB + fro);
        };
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int lr = 16 * (w * MT + m) + l15;
            i32x4 af[NX];
#pragma unroll
            for (int s3 = 0; s3 < NX; ++s3)
                af[s3] = *reinterpret_cast<const i32x4 *>(Xs + s3 * B6_XBYTES + 16 * (w * MT + m) * B6_LDB + fro);
            if (!DROP) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const i32x4 b0 = bfrag(t, 0), b1 = bfrag(t, 1), b2 = bfrag(t, 2);
                    f32x4 c = acc[m][t][0];
                    if (!XBF) {
                        c = b6_mfma(af[1], b1, c);      // small terms first
                        c = b6_mfma(af[2], b0, c);
                        c = b6_mfma(af[1], b0, c);
                    }
                    c = b6_mfma(af[0], b2, c);
                    c = b6_mfma(af[0], b1, c);
                    c = b6_mfma(af[0], b0, c);
                    acc[m][t][0] = c;
                }
            } else {
                // keep masks of this lane's 8 elements (row nglob, k = k0 + 8*l4 + j), four heads at a time:
                // call c serves heads 4c..4c+3: x = fields of heads 4c, 4c+1; y = heads 4c+2, 4c+3
                const uint32_t nglob = (uint32_t)(row0 + lr + a.row_offset);
                uint32_t kw[2] = {0u, 0u};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    uint32_t mk[8][2];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t kglob = (uint32_t)(k0 + 8 * l4 + j);
                        const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, HAN_STREAM_SEQ, nglob, kglob * 2u + (uint32_t)c);
                        mk[j][0] = b6_keep_masks(rn.x, thr2, one2);
                        mk[j][1] = b6_keep_masks(rn.y, thr2, one2);
                        if constexpr (KEEP) {
                            // this lane's 8 elements are one octet of its row: 64 keep bits (8 elements x 8 heads) in
                            // the bit order of the dW kernel's A-operand lane mask (han_hip.h): element j = 4q + i,
                            // head k = 4c + 2wd + half -> bit 32q + 16*half + 4*(2c + wd) + i; a mask word holds the
                            // two halves (heads 4c+2wd, 4c+2wd+1) in its low / high 16 bits: one and-or per word
#pragma unroll
                            for (int wd = 0; wd < 2; ++wd)
                                kw[j >> 2] |= mk[j][wd] & (0x00010001u << (4 * (2 * c + wd) + (j & 3)));
                        }
                    }
#pragma unroll
                    for (int hq = 0; hq < 4; ++hq) {
                        const int head = 4 * c + hq;
                        const int t = head >> 1, hh = head & 1, wd = hq >> 1;
                        const uint32_t sel = (head & 1) ? 0x07060302u : 0x05040100u;
                        i32x4 am[NX];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t pm = __builtin_amdgcn_perm(mk[2 * i + 1][wd], mk[2 * i][wd], sel);
#pragma unroll
                            for (int s3 = 0; s3 < NX; ++s3) am[s3][i] = af[s3][i] & (int)pm;
                        }
                        const i32x4 b0 = bfrag(t, 0), b1 = bfrag(t, 1), b2 = bfrag(t, 2);
                        f32x4 cc = acc[m][t][hh];
                        if (!XBF) {
                            cc = b6_mfma(am[1], b1, cc);
                            cc = b6_mfma(am[2], b0, cc);
                            cc = b6_mfma(am[1], b0, cc);
                        }
                        cc = b6_mfma(am[0], b2, cc);
                        cc = b6_mfma(am[0], b1, cc);
                        cc = b6_mfma(am[0], b0, cc);
                        acc[m][t][hh] = cc;
                    }
                }
                // 16 contiguous lanes hold the same 8-byte piece (k0 & 96) / 8 + l4 of 16 consecutive rows: XOR-ing the
                // piece index with the row spreads them over 16 different pieces = 32 banks (un-swizzled: 16-way)
                if constexpr (KEEP)
                    *reinterpret_cast<uint2 *>(Kacc + lr * 128 + 8 * ((((k0 & 96) >> 3) + l4) ^ (lr & 15))) = make_uint2(kw[0], kw[1]);
            }
        }
    }
    if constexpr (KEEP) {
        __syncthreads();
        flush_keep(((a.F - 1) >> 7) << 7);      // the last (possibly partial) group of K-steps
    }
    // epilogue: C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg (as the fp32 kernel's)
    const int myhh = HPT > 1 ? l15 / 8 : 0;
    float a1c[4], a2c[4];
    if (DROP && a.f1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a1c[t] = a.a1[16 * t + l15];
            a2c[t] = a.a2[16 * t + l15];
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 16 * (w * MT + m) + l4 * 4 + r;
            float vst[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = acc[m][t][0][r];
                if (HPT > 1) v = myhh ? acc[m][t][HPT - 1][r] : v;
                if (DROP) v *= a.inv_keep_in;
                uint32_t keepbit = 0, stamp = 0;
                if (a.thr_fts < HAN_KEEP_ALL) {
                    const int d = 16 * t + l15;
                    const HanRand64 rn = han_rand64(a.seed_lo, a.seed_hi, a.fts_stream,
                                                    (uint32_t)(row + a.row_offset), (uint32_t)(d >> 2));
                    keepbit = rn.field(d & 3) < a.thr_fts ? 1u : 0u;
                    stamp = 1;
                }
                if (a.h_bf16) {
                    uint32_t b = han_f32_to_bf16_bits(v);
                    if (stamp) b = (b & ~1u) | keepbit;
                    if (row < a.N) reinterpret_cast<uint16_t *>(a.H)[row * HAN_D + 16 * t + l15] = (uint16_t)b;
                    vst[t] = __uint_as_float(b << 16);
                } else {
                    if (stamp) v = __uint_as_float((__float_as_uint(v) & ~1u) | keepbit);
                    if (row < a.N) reinterpret_cast<float *>(a.H)[row * HAN_D + 16 * t + l15] = v;
                    vst[t] = v;
                }
            }
            if (DROP && a.f1) tile_scores<8>(a, row, row < a.N, vst, a1c, a2c, l15);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Eval forward of SEVERAL meta-paths that share one feature matrix (the reference feeds ONE matrix to every
// meta-path: ex_acm3025.py:86, models/gat.py:39), on the bf16 x 6 matrix pipe: the X tile is read from HBM,
// split into its three bf16 terms and staged ONCE for NP x 64 output columns (round 2 read, split and staged
// it once per meta-path: 4 launches of 0.40 ms at SYN-1M, each bound by the staging of a tile that feeds
// only 64 columns).  8 waves x one 16-row tile, all NP x 4 column tiles per wave; W_p tiles side by side in LDS.
// blockIdx.y = group of NP meta-paths.  Scores (F' = 8) in the epilogue from the rows as stored.
// ---------------------------------------------------------------------------------------------
struct ProjMultiArgs {
    const void *X;
    int64_t ldx;
    const float *W;          // (P, F, 64)
    const unsigned char *wimg;   // (P, ktiles, B6_WTILE): the pre-split LDS-ready images (project_wimage_kernel)
    void *H;                 // (P, N, 64) fp32 or bf16
    int h_bf16;
    int64_t N;
    int F;
    int p_first;             // first meta-path of this launch
    int P;                   // meta-paths in the image (its layout is [K-step][meta-path])
    const float *a1, *a2, *b1, *b2;     // (P, K, FP), (P, K)
    float *f1, *f2;          // (P, N, K)
    int K;
    int fuse_scores;         // F' == 8: scores in the epilogue
};

// MT = 16-row tiles per wave: a B fragment read from LDS serves MT row tiles (at MT = 1 the 8 waves re-read the whole W
// image for 16 rows each and the kernel is bound by LDS bandwidth: 1.37 ms for four meta-paths at SYN-1M)
template <bool XBF, int NP, int MT>
__global__ __launch_bounds__(512) void project_fwd_b6_multi_kernel(const ProjMultiArgs a) {
    constexpr int NX = XBF ? 1 : 3;
    constexpr int NT = 512;
    constexpr int WU = NP * (B6_WTILE / 8) / NT; // 8-byte pieces of the W images per thread per K-step
    constexpr int ROWS = 128 * MT;               // rows per block
    constexpr int XB = ROWS * B6_LDB;            // bytes of one term's X image
    extern __shared__ __attribute__((aligned(16))) unsigned char mlds[];
    unsigned char *Xs = mlds;                    // [NX][ROWS][64 B, slots swizzled: b6_off]
    unsigned char *Ws = mlds + NX * XB;          // [NP][3][64][64 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int p0 = a.p_first + (int)blockIdx.y * NP;
    const int ktiles = (a.F + 31) >> 5;
    // the block's W tile of a K-step = the NP consecutive images (p0 .. p0 + NP - 1) of that K-step: the image is laid out
    // [K-step][meta-path], and the LDS copy keeps that order ([pi][term][column][64 B]) -- one contiguous run of 8-byte pieces
    const unsigned char *wsrc = a.wimg + (int64_t)p0 * B6_WTILE + 8 * tid;

    f32x4 acc[MT][NP * 4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NP * 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4_t xr[2 * MT];
    uint2 wq[WU];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const int idx = tid + NT * i;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            const int64_t row = row0 + r;
            const int64_t rc = row < a.N ? row : a.N - 1;
            const int kc = k0 + c4 < a.F ? k0 + c4 : a.F - 4;
            xr[i] = load_x4(a.X, XBF, rc * a.ldx + kc);      // raw; zeroed when staged (no use before the MFMA phase)
        }
        const int kt = (k0 >> 5) < ktiles ? (k0 >> 5) : ktiles - 1;
#pragma unroll
        for (int i = 0; i < WU; ++i)
            wq[i] = *reinterpret_cast<const uint2 *>(wsrc + (int64_t)kt * a.P * B6_WTILE + 8 * NT * i);
    };
    load_tile(0);
    for (int k0 = 0; k0 < a.F; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const int idx = tid + NT * i;
            const int off = b6_off(idx >> 3, (idx & 7) >> 1) + (idx & 1) * 8;
            const bool ok = row0 + (idx >> 3) < a.N && k0 + (idx & 7) * 4 < a.F;
            uint32_t h[4], m[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) b6_split(ok ? xr[i][e] : 0.f, h[e], m[e], l[e]);
            *reinterpret_cast<uint2 *>(Xs + off) = make_uint2(b6_pack(h[0], h[1]), b6_pack(h[2], h[3]));
            if (!XBF) {
                *reinterpret_cast<uint2 *>(Xs + XB + off) = make_uint2(b6_pack(m[0], m[1]), b6_pack(m[2], m[3]));
                *reinterpret_cast<uint2 *>(Xs + 2 * XB + off) = make_uint2(b6_pack(l[0], l[1]), b6_pack(l[2], l[3]));
            }
        }
#pragma unroll
        for (int i = 0; i < WU; ++i) *reinterpret_cast<uint2 *>(Ws + 8 * (tid + NT * i)) = wq[i];      // straight copy of the images' K-step
        __syncthreads();
        load_tile(k0 + 32);      // unconditional, clamped: see project_fwd_b6_kernel
        const int fro = l15 * B6_LDB + ((l4 ^ b6_swz(l15)) << 4);      // this lane's fragment inside a 16-row tile
        i32x4 af[MT][NX];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s3 = 0; s3 < NX; ++s3)
                af[m][s3] = *reinterpret_cast<const i32x4 *>(Xs + s3 * XB + 16 * (w * MT + m) * B6_LDB + fro);
#pragma unroll
        for (int t = 0; t < NP * 4; ++t) {
            const unsigned char *bp = Ws + (t >> 2) * B6_WTILE + 16 * (t & 3) * B6_LDB + fro;
            const i32x4 b0 = *reinterpret_cast<const i32x4 *>(bp);
            const i32x4 b1 = *reinterpret_cast<const i32x4 *>(bp + B6_WBYTES);
            const i32x4 b2 = *reinterpret_cast<const i32x4 *>(bp + 2 * B6_WBYTES);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 c = acc[m][t];
                if (!XBF) {
                    c = b6_mfma(af[m][1], b1, c);      // small terms first
                    c = b6_mfma(af[m][2], b0, c);
                    c = b6_mfma(af[m][1], b0, c);
                }
                c = b6_mfma(af[m][0], b2, c);
                c = b6_mfma(af[m][0], b1, c);
                c = b6_mfma(af[m][0], b0, c);
                acc[m][t] = c;
            }
        }
    }
    // epilogue: C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
        const int p = p0 + pi;
        ProjFwdArgs pa;       // what tile_scores reads
        pa.f1 = a.f1 + (int64_t)p * a.N * a.K; pa.f2 = a.f2 + (int64_t)p * a.N * a.K;
        pa.b1 = a.b1 + p * a.K; pa.b2 = a.b2 + p * a.K;
        float a1c[4], a2c[4];
        if (a.fuse_scores) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a1c[t] = a.a1[p * HAN_D + 16 * t + l15];
                a2c[t] = a.a2[p * HAN_D + 16 * t + l15];
            }
        }
#pragma unroll
        for (int mr = 0; mr < MT * 4; ++mr) {
            const int m = mr >> 2, r = mr & 3;
            const int64_t row = row0 + 16 * (w * MT + m) + l4 * 4 + r;
            float vst[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = acc[m][pi * 4 + t][r];
                const int64_t o = ((int64_t)p * a.N + row) * HAN_D + 16 * t + l15;
                if (a.h_bf16) {
                    const uint32_t b = han_f32_to_bf16_bits(v);
                    if (row < a.N) reinterpret_cast<uint16_t *>(a.H)[o] = (uint16_t)b;
                    vst[t] = __uint_as_float(b << 16);
                } else {
                    if (row < a.N) reinterpret_cast<float *>(a.H)[o] = v;
                    vst[t] = v;
                }
            }
            if (a.fuse_scores) tile_scores<8>(pa, row, row < a.N, vst, a1c, a2c, l15);
        }
    }
}

// split-F epilogue: H[row] = sum over the f-chunks (fixed order), then the same keep-bit
// stamping / bf16 rounding as the unsplit kernel's epilogue -- and the attention scores f1 / f2
// (layers.py:23-24) from the row exactly as stored, in the same pass (round 3: one launch per
// projection less on the short inputs, whose epochs are launch-bound).
template <int FP, bool BF>
__global__ __launch_bounds__(256) void project_finish_kernel(const ProjFwdArgs a_in, int nsplit, float *f1, float *f2) {
    ProjFwdArgs a = a_in;
    han_resolve_seed(a.seed_lo, a.seed_hi, a.seed_dev);
    constexpr int K = HAN_D / FP;
    const int q = threadIdx.x & 15;
    const int head = (4 * q) / FP;
    const int64_t grp0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t ngrp = (int64_t)gridDim.x * 16;
    const float4_t a14 = *reinterpret_cast<const float4_t *>(a.a1 + 4 * q);
    const float4_t a24 = *reinterpret_cast<const float4_t *>(a.a2 + 4 * q);
    const float b1 = a.b1[head], b2 = a.b2[head];
    // all 16 lanes of a group run the same trip count -> the in-head shuffles are safe
    for (int64_t row = grp0; row < a.N; row += ngrp) {
        float4_t v = {0.f, 0.f, 0.f, 0.f};
        for (int sidx = 0; sidx < nsplit; ++sidx) {
            const float4_t pv = *reinterpret_cast<const float4_t *>(a.partial + ((int64_t)sidx * a.N + row) * HAN_D + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += pv[e];
        }
        HanRand64 rn = {0u, 0u};
        if (a.thr_fts < HAN_KEEP_ALL)      // layers.py:31-32, d = 4q + e -> counter d/4 = q, field e
            rn = han_rand64(a.seed_lo, a.seed_hi, a.fts_stream, (uint32_t)(row + a.row_offset), (uint32_t)q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t keepbit = rn.field(e) < a.thr_fts ? 1u : 0u;
            if (BF) {
                uint32_t b = han_f32_to_bf16_bits(v[e]);
                if (a.thr_fts < HAN_KEEP_ALL) b = (b & ~1u) | keepbit;
                v[e] = __uint_as_float(b << 16);          // the value as stored (what the scores see)
            } else if (a.thr_fts < HAN_KEEP_ALL) {
                v[e] = __uint_as_float((__float_as_uint(v[e]) & ~1u) | keepbit);
            }
        }
        if (BF) {      // already rounded: pack the high halves
            uint2 w;
            w.x = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xFFFF0000u);
            w.y = (__float_as_uint(v[2]) >> 16) | (__float_as_uint(v[3]) & 0xFFFF0000u);
            *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(a.H) + row * 64 + 4 * q) = w;
        } else {
            *reinterpret_cast<float4_t *>(reinterpret_cast<float *>(a.H) + row * 64 + 4 * q) = v;
        }
        float s1 = v[0] * a14[0] + v[1] * a14[1] + v[2] * a14[2] + v[3] * a14[3];
        float s2 = v[0] * a24[0] + v[1] * a24[1] + v[2] * a24[2] + v[3] * a24[3];
#pragma unroll
        for (int o = 1; o < FP / 4; o <<= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        if ((4 * q) % FP == 0) {
            f1[row * K + head] = s1 + b1;
            f2[row * K + head] = s2 + b2;
        }
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
                if (row < n_end && f0 + c4 < a.F) v = load_x4_tail(a.X, a.x_bf16, row * a.ldx + f0 + c4, a.F - (f0 + c4));
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

// ---------------------------------------------------------------------------------------------
// dW on v_mfma_f32_4x4x1_16B_f32 with the keep table as the A operand's lane mask.
//
// The 16x16x4 kernel above pays the per-head input dropout twice: it regenerates every (row, feature,
// head) draw (6x the VALU instructions of the kernel without dropout, profiles/r02_pmc_k1_b6.json) and it
// issues every 16-column tile once per head it covers, discarding half of each product.  The 16-block
// 4x4x1 form has neither problem: one instruction computes 16 independent 4x4 outer products
//     D_b[i][j] += A_b[i] * B_b[j]          (b = lane / 4; A: i = lane % 4; B and D: j = lane % 4, D reg = i;
//                                            map and 8-cycle issue measured with tools/micro/ubench.hip)
// so a block can carry ONE head: block b = (q, b') takes the 4 features f = oct + 4q + i (q = lane / 32) of
// head k(b') = 2 (b' % 4) + b' / 4 against 4 of that head's columns.  The A operand of lane l is then
// X[n][oct + 4q + i] if bit l of the row's 64-bit keep word is set (han_hip.h: the forward wrote the table
// in exactly this bit order): one v_bfe_i32 + one v_and_b32, no hash, no wasted column, and each masked A
// register serves two MFMAs (column quads 0-3 and 4-7 of every head).  Same fp32 pipe, same 64 flop / clk /
// SIMD, but 1024 useful flop per 16 cycles of issue instead of 1024 useful out of 2048 per 32.
// Block: 128 features (4 waves x 4 octets) x all 64 columns over a chunk of rows; X, dH and the keep words
// of 32 rows are staged through LDS one tile ahead, laid out so that a row costs a wave three LDS reads:
//   Xs[n][wave][q][i][octet]      one ds_read_b128 = the lane's 4 A values
//   Gs[n][b'][j][parity]          one ds_read_b64  = its 2 B values
//   Ks[n][wave][q][octet]         one ds_read_b128 = the 4 half keep words holding its bits
// Partial tiles go to a slab (fixed-order second stage, no float atomics).
// ---------------------------------------------------------------------------------------------
constexpr int DW_BN = 32;            // rows per staged tile
constexpr int DW_FB = 128;           // dW rows (features) per block

struct ProjBwdBlkArgs {
    const void *X;
    int64_t ldx;
    const float *dH;
    const uint8_t *keep;    // N rows of F bytes (+ slack), or null when DROP is false
    float *slab;            // [nchunks][F][64]
    int64_t N;
    int F;
    int64_t rows_per_chunk;
    float inv_keep_in;
};

template <bool DROP, bool XBF>
__global__ __launch_bounds__(256, 4) void project_bwd_blk_kernel(const ProjBwdBlkArgs a) {
    __shared__ __attribute__((aligned(16))) float Xs[DW_BN * DW_FB];
    __shared__ __attribute__((aligned(16))) float Gs[DW_BN * HAN_D];
    __shared__ __attribute__((aligned(16))) uint32_t Ks[DROP ? DW_BN * 32 : 4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int f0 = blockIdx.x * DW_FB;
    const int64_t chunk = blockIdx.y;
    const int64_t n_begin = chunk * a.rows_per_chunk;
    const int64_t n_end = (n_begin + a.rows_per_chunk < a.N) ? n_begin + a.rows_per_chunk : a.N;
    const int q = lane >> 5, bq = (lane & 31) >> 2, i4 = lane & 3;

    f32x4 acc[4][2];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[o][h] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging registers: X tile 32 x 128 (4 x 16 B per thread), dH tile 32 x 64 (2 x 16 B), keep tile 32 x 128 B (2 x 8 B).
    // Buffer loads: the range check of the descriptor returns 0 for rows beyond the chunk, a thread whose
    // columns lie beyond F carries an out-of-range offset for good, and the step advance is a scalar offset --
    // the tile costs no vector ALU work (the selects and 64-bit address arithmetic of plain loads were 58 % of
    // this kernel's vector instructions, and a 2-pass MFMA does not share its issue cycles with them:
    // profiles/r03_pmc_k1.json)
    constexpr int XE = XBF ? 2 : 4;                    // bytes per X element
    const int64_t nrows = n_end - n_begin;
    const uint32_t OOB = 0x7FFFFFF0u;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(reinterpret_cast<const char *>(a.X) + (n_begin * a.ldx + f0) * XE), 0,
        (int)(((nrows - 1) * a.ldx + (a.F - f0 < DW_FB ? a.F - f0 : DW_FB)) * XE), 0x00020000);
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void *)(a.dH + n_begin * HAN_D), 0, (int)(nrows * HAN_D * 4), 0x00020000);
    const auto rs_k = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(DROP ? a.keep + n_begin * (int64_t)a.F + f0 : (const uint8_t *)a.X), 0,
        DROP ? (int)((nrows - 1) * a.F + (a.F - f0 < DW_FB ? a.F - f0 : DW_FB)) : 0, 0x00020000);
    const int xc4 = (tid & 31) * 4;
    const uint32_t vx = f0 + xc4 < a.F ? (uint32_t)(((tid >> 5) * a.ldx + xc4) * XE) : OOB;      // + 8 i rows
    const uint32_t vg = (uint32_t)((tid >> 3) * (HAN_D * 4) + (tid & 7) * 32);                // (row, head): 8 columns = 2 x 16 B
    const uint32_t vk = f0 + 16 * (tid & 7) < a.F ? (uint32_t)((tid >> 3) * a.F + 16 * (tid & 7)) : OOB;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    float4_t xr[4], gr[2];
    u32x2 kr[2];
    // The whole offset travels in the VECTOR offset: the descriptor's range check covers inst_offset + voffset only
    // (an SGPR offset is added after the check), and the check is what zeroes the rows beyond the chunk.
    auto load_tile = [&](int64_t n0) {
        const uint32_t step = (uint32_t)(n0 - n_begin);
        const uint32_t in_x = vx == OOB ? 0u : 1u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t vo = in_x ? vx + (step + 8 * i) * (uint32_t)(a.ldx * XE) : OOB;
            if (XBF) {
                const u32x2 wv = __builtin_amdgcn_raw_buffer_load_b64(rs_x, vo, 0, 0);
                xr[i][0] = __uint_as_float(wv.x << 16);
                xr[i][1] = __uint_as_float(wv.x & 0xFFFF0000u);
                xr[i][2] = __uint_as_float(wv.y << 16);
                xr[i][3] = __uint_as_float(wv.y & 0xFFFF0000u);
            } else {
                const u32x4 wv = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, 0, 0);
                xr[i] = __builtin_bit_cast(float4_t, wv);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)      // the 8 columns of this thread's (row, head): parity 0 and parity 1 quads
            gr[i] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_g, vg + step * (HAN_D * 4) + 16 * i, 0, 0));
        if (DROP) {      // features beyond F meet X == 0, so their (arbitrary) keep bits do not matter
            const uint32_t vo = vk == OOB ? OOB : vk + step * (uint32_t)a.F;
            kr[0] = __builtin_amdgcn_raw_buffer_load_b64(rs_k, vo, 0, 0);
            kr[1] = __builtin_amdgcn_raw_buffer_load_b64(rs_k, vo == OOB ? OOB : vo + 8, 0, 0);
        }
    };
    // dword offsets of this lane's operands inside a row of the three LDS images
    // (the wave's 32 dwords of a row are rotated by 4 w: with it the staging stores of the four waves' features fall
    //  on 32 different banks -- without, 4-way conflicts on every store made LDS co-critical with the matrix pipe)
    const int a_off = 32 * w + ((16 * q + 4 * i4 + 4 * w) & 31);
    const int b_off = 8 * bq + 2 * i4;
    const int k_off = 8 * w + 4 * q;
    const uint32_t my_bit = (uint32_t)(lane & 31);

    load_tile(n_begin);
    for (int64_t n0 = n_begin; n0 < n_end; n0 += DW_BN) {
        __syncthreads();      // the previous tile's operand reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 5, k = idx & 31;      // features 4k .. 4k+3: wave k/8, octet (k%8)/2, half k%2, i = e
            const int wv = k >> 3;
            float *dst = Xs + r * DW_FB + 32 * wv + ((k & 7) >> 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(16 * (k & 1) + 4 * e + 4 * wv) & 31] = xr[i][e];
        }
        {      // dH: this thread's (row, head) as [j][parity] pairs -- 32 contiguous bytes, two 16-B stores
            const int hd = tid & 7;
            const int rank = (hd & 1) * 4 + (hd >> 1);                 // b' of that head
            float *dst = Gs + (tid >> 3) * HAN_D + 8 * rank;
            *reinterpret_cast<float4_t *>(dst) = (float4_t){gr[0][0], gr[1][0], gr[0][1], gr[1][1]};
            *reinterpret_cast<float4_t *>(dst + 4) = (float4_t){gr[0][2], gr[1][2], gr[0][3], gr[1][3]};
        }
        if (DROP) {
            const int seg = tid & 7;                                   // octets 2 seg, 2 seg + 1 of the tile's 16
            uint32_t *dst = Ks + (tid >> 3) * 32 + 8 * (seg >> 1) + 2 * (seg & 1);
            *reinterpret_cast<uint2 *>(dst) = make_uint2(kr[0].x, kr[1].x);         // q = 0 halves of the two octets
            *reinterpret_cast<uint2 *>(dst + 4) = make_uint2(kr[0].y, kr[1].y);     // q = 1 halves
        }
        __syncthreads();
        load_tile(n0 + DW_BN);     // in flight under the MFMAs (unconditional: beyond the chunk the descriptor returns 0)
        // Three-stage software pipeline over the 32 rows of the tile, spelled out because hipcc's own schedule
        // (reads of two rows, then a wait right behind them, one temporary through every mask / and / MFMA chain)
        // exposes the LDS latency every other row: row n + 2 is being read, row n + 1's A operands are being
        // masked, row n's eight MFMAs issue; sched_group_barrier pins that interleave (2 MFMA : 2 VALU).
        float4_t xa[3];
        float2 bb[3];
        uint4 kk[3];
        auto fetch = [&](int n, int slot) {
            xa[slot] = *reinterpret_cast<const float4_t *>(Xs + n * DW_FB + a_off);
            bb[slot] = *reinterpret_cast<const float2 *>(Gs + n * HAN_D + b_off);
            if (DROP) kk[slot] = *reinterpret_cast<const uint4 *>(Ks + n * 32 + k_off);
        };
        auto masked = [&](int slot, int o) -> float {
            if (!DROP) return xa[slot][o];
            const uint32_t kwd = o == 0 ? kk[slot].x : o == 1 ? kk[slot].y : o == 2 ? kk[slot].z : kk[slot].w;
            return __int_as_float(__float_as_int(xa[slot][o]) & __builtin_amdgcn_sbfe((int)kwd, my_bit, 1u));
        };
        float am[2][4];
        fetch(0, 0);
        fetch(1, 1);
#pragma unroll
        for (int o = 0; o < 4; ++o) am[0][o] = masked(0, o);
#pragma unroll
        for (int n = 0; n < DW_BN; ++n) {
            const int cur = n & 1;
            if (n + 2 < DW_BN) fetch(n + 2, (n + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);      // the reads of row n + 2 go out BEFORE row n's MFMAs, and stay there
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                acc[o][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(am[cur][o], bb[n % 3].x, acc[o][0], 0, 0, 0);
                acc[o][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(am[cur][o], bb[n % 3].y, acc[o][1], 0, 0, 0);
                if (n + 1 < DW_BN) am[cur ^ 1][o] = masked((n + 1) % 3, o);
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                              // 2 MFMA
                if (DROP && n + 1 < DW_BN) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // 2 VALU (bfe, and)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D: lane = 4 b + j, register = i -> dW[f = f0 + 32 w + 8 o + 4 q + i][col = 8 head(b') + 4 parity + j]
    float *out = a.slab + chunk * (int64_t)a.F * HAN_D;
    const int head = 2 * (bq & 3) + (bq >> 2);
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = f0 + 32 * w + 8 * o + 4 * q + r;
                const float v = acc[o][h][r];
                if (f < a.F) out[(int64_t)f * HAN_D + 8 * head + 4 * h + i4] = DROP ? v * a.inv_keep_in : v;
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

// bytes of the pre-split W image of P meta-paths (project_wimage_kernel)
static size_t wimage_bytes(int F, int P) { return (size_t)((F + 31) / 32) * (size_t)P * B6_WTILE; }
// shapes the bf16 x 6 matrix-pipe kernels may run on (whole-F blocks of 128 rows)
static bool b6_shape(int64_t N, int nsplit) { return nsplit == 1 && N >= 64 * 256; }

extern "C" size_t han_project_fwd_multi_workspace(int64_t N, int F, int K, int FP, int P) {
    (void)K; (void)FP;
    int mt, nsplit, f_chunk;
    fwd_geometry(N > 0 ? N : 0, F, &mt, &nsplit, &f_chunk);
    if (nsplit > 1) return (size_t)nsplit * (size_t)N * HAN_D * sizeof(float);      // split-F partial tiles (one meta-path at a time)
    return b6_shape(N, nsplit) ? wimage_bytes(F, P > 0 ? P : 1) : 0;
}

extern "C" size_t han_project_fwd_workspace(int64_t N, int F, int K, int FP) {
    return han_project_fwd_multi_workspace(N, F, K, FP, 1);
}

// the shapes for which the training forward writes (and dW reads) the keep table: the reference head shape on the
// matrix-pipe forward (whole-F blocks of 128 rows, no split-F: at least 256 row tiles)
static bool keep_table_shape(int64_t N, int F, int64_t ldx, int K, int FP) {
    return K == 8 && FP == 8 && F % 8 == 0 && ldx % 4 == 0 && N >= 128 * 256;
}

extern "C" size_t han_project_keep_bytes(int64_t N, int F, int64_t ldx, int K, int FP) {
    if (N <= 0 || F <= 0 || !keep_table_shape(N, F, ldx, K, FP)) return 0;
    return (size_t)N * (size_t)F + 128;
}

extern "C" int han_project_fwd(const void *X, int x_dtype, int64_t ldx, const float *W, const float *a1,
                               const float *a2, const float *b1, const float *b2, void *H, int table_dtype,
                               float *f1, float *f2, void *workspace, size_t workspace_bytes, int64_t N, int F,
                               int K, int FP, float in_drop, float fts_drop, uint64_t seed,
                               const uint64_t *seed_dev, int64_t row_offset, uint8_t *keep, int flags,
                               void *stream) {
    if (N == 0) return 0;   // nothing to do; row pointers of empty tensors may be null
    if (!X || !W || !a1 || !a2 || !b1 || !b2 || !H || !f1 || !f2 || N < 0 || F <= 0 || ldx < F)
        return HAN_E_BADARG;
    if (!fp_ok(K, FP)) return HAN_E_UNSUPPORTED;
    if ((x_dtype != HAN_DTYPE_F32 && x_dtype != HAN_DTYPE_BF16) ||
        (table_dtype != HAN_DTYPE_F32 && table_dtype != HAN_DTYPE_BF16))
        return HAN_E_UNSUPPORTED;
    if (in_drop < 0.f || in_drop >= 1.f || fts_drop < 0.f || fts_drop >= 1.f) return HAN_E_BADARG;
    if ((flags & HAN_FLAG_K1_EXACT_PIPE) && (flags & HAN_FLAG_K1_MATRIX_PIPE)) return HAN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    ProjFwdArgs a;
    a.X = X; a.ldx = ldx; a.W = W; a.H = H; a.N = N; a.F = F;
    a.x_bf16 = x_dtype == HAN_DTYPE_BF16; a.h_bf16 = table_dtype == HAN_DTYPE_BF16;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_dev = seed_dev;
    a.thr_in = in_drop > 0.f ? han_keep_threshold(1.f - in_drop) : HAN_KEEP_ALL;
    a.thr_fts = fts_drop > 0.f ? han_keep_threshold(1.f - fts_drop) : HAN_KEEP_ALL;
    a.fts_stream = HAN_STREAM_FTS + 4u * (uint32_t)HAN_FLAG_FTS_SLICE_OF(flags);
    a.inv_keep_in = 1.f / (1.f - in_drop);
    a.row_offset = row_offset;
    a.keep = in_drop > 0.f ? keep : nullptr;
    const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)X & (a.x_bf16 ? 7 : 15)) == 0);
    const bool vec32 = true;      // the exact-fp32 kernels: quad loads at element alignment, element loads for a partial last quad
    int mt, nsplit;
    fwd_geometry(N, F, &mt, &nsplit, &a.f_chunk);
    a.partial = nullptr;
    a.a1 = a1; a.a2 = a2; a.b1 = b1; a.b2 = b2; a.f1 = f1; a.f2 = f2;      // scores fused into the epilogue ...
    if (nsplit > 1) {
        a.f1 = nullptr; a.f2 = nullptr;                                     // ... except on the split-F path
        if (!workspace || workspace_bytes < han_project_fwd_workspace(N, F, K, FP)) return HAN_E_WORKSPACE;
        a.partial = (float *)workspace;
    }
    // bf16 x 6 matrix-pipe kernel (fp32-class accuracy, see project_fwd_b6_kernel): whole-F blocks of 128
    // rows with 16-byte X loads; with dropout it is built for the reference head shape (8 x 8).
    // Used for the training forward (measured at SYN-1M in one process: 0.77 ms against 0.96 ms for the
    // exact-fp32 kernel); without dropout both take the same time (0.48 / 0.50 ms: staging-latency bound), so
    // the eval forward stays on the exact-fp32 pipe unless HAN_FLAG_K1_MATRIX_PIPE asks for this kernel.  (A
    // wave-local variant -- no LDS staging of X, no barrier per K-step, W chunks of 128 k in LDS -- measured the
    // same: 0.49 / 0.80 ms against 0.46 / 0.78 ms for this one, eval / training, kernel_bench.py k1.)
    // bf16 features are their own high term (no split, three products instead of six): there the matrix-pipe
    // kernel is also the faster eval forward
    const bool b6_want = (in_drop > 0.f || a.x_bf16) ? !(flags & HAN_FLAG_K1_EXACT_PIPE) : (flags & HAN_FLAG_K1_MATRIX_PIPE) != 0;
    const bool b6 = vec && b6_shape(N, nsplit) && (in_drop == 0.f || (K == 8 && FP == 8)) && b6_want;
    if (a.keep && !(b6 && keep_table_shape(N, F, ldx, K, FP))) return HAN_E_BADARG;   // no kernel writes a table here
    a.wimg = nullptr;
    if (b6) {
        // W is split into its three bf16 terms ONCE, into the LDS-ready image every block copies per K-step
        if (!workspace || workspace_bytes < wimage_bytes(F, 1)) return HAN_E_WORKSPACE;
        const int ktiles = (F + 31) / 32;
        project_wimage_kernel<<<(ktiles * 512 + 255) / 256, 256, 0, st>>>(W, (unsigned char *)workspace, F, ktiles, 1);
        HAN_CHECK_LAUNCH();
        a.wimg = (const unsigned char *)workspace;
        const dim3 g6((unsigned)((N + B6_ROWS - 1) / B6_ROWS));
#define HAN_LAUNCH_B6(D_, X_, K_)                                                               \
    do {                                                                                        \
        if (flags & HAN_FLAG_K1_4WAVE) project_fwd_b6_kernel<D_, X_, K_, 4><<<g6, 256, 0, st>>>(a); \
        else project_fwd_b6_kernel<D_, X_, K_, 8><<<g6, 512, 0, st>>>(a);                       \
    } while (0)
        if (in_drop > 0.f) {
            if (a.keep) {
                if (a.x_bf16) HAN_LAUNCH_B6(true, true, true);
                else HAN_LAUNCH_B6(true, false, true);
            } else {
                if (a.x_bf16) HAN_LAUNCH_B6(true, true, false);
                else HAN_LAUNCH_B6(true, false, false);
            }
        } else {
            if (a.x_bf16) HAN_LAUNCH_B6(false, true, false);
            else HAN_LAUNCH_B6(false, false, false);
        }
#undef HAN_LAUNCH_B6
        HAN_CHECK_LAUNCH();
    }
    const dim3 grid((unsigned)((N + 64 * mt - 1) / (64 * mt)), (unsigned)nsplit);
#define HAN_LAUNCH_FWD(MTC)                                                                  \
    HAN_DISPATCH_FP(FP, {                                                                    \
        if (in_drop > 0.f) {                                                                 \
            if (vec32) project_fwd_kernel<FPC, true, MTC, true><<<grid, 256, 0, st>>>(a);    \
            else project_fwd_kernel<FPC, true, MTC, false><<<grid, 256, 0, st>>>(a);         \
        } else {                                                                             \
            if (vec32) project_fwd_kernel<FPC, false, MTC, true><<<grid, 256, 0, st>>>(a);   \
            else project_fwd_kernel<FPC, false, MTC, false><<<grid, 256, 0, st>>>(a);        \
        }                                                                                    \
    })
    if (b6) { /* done above */ } else if (mt == 2) { HAN_LAUNCH_FWD(2) } else { HAN_LAUNCH_FWD(1) }
#undef HAN_LAUNCH_FWD
    HAN_CHECK_LAUNCH();
    if (nsplit > 1) {      // sums the partial tiles, stamps / rounds, and takes the scores from the stored row
        const int fgrid = han_grid_for(N, 16, 256 * 8);
        a.a1 = a1; a.a2 = a2; a.b1 = b1; a.b2 = b2;
        if (a.h_bf16) { HAN_DISPATCH_FP(FP, { project_finish_kernel<FPC, true><<<fgrid, 256, 0, st>>>(a, nsplit, f1, f2); }) }
        else { HAN_DISPATCH_FP(FP, { project_finish_kernel<FPC, false><<<fgrid, 256, 0, st>>>(a, nsplit, f1, f2); }) }
        HAN_CHECK_LAUNCH();
        return 0;
    }
    if (a.f1 && !(b6 && in_drop == 0.f)) return 0;      // f1 / f2 were written by the epilogue
    ScoreArgs s;
    s.H = H; s.a1 = a1; s.a2 = a2; s.b1 = b1; s.b2 = b2; s.f1 = f1; s.f2 = f2;
    s.N = N;
    const int sgrid = han_grid_for(N, 16, 256 * 8);
    // the lane map of the scores follows the head width for both storage types (a bf16 table with F' != 8 through the
    // 8 x 8 instantiation wrote f1 / f2 out of bounds for K < 8 and a wrong layout for K = 16)
    if (a.h_bf16) {
        HAN_DISPATCH_FP(FP, { project_scores_kernel<FPC, true><<<sgrid, 256, 0, st>>>(s); })
    } else {
        HAN_DISPATCH_FP(FP, { project_scores_kernel<FPC, false><<<sgrid, 256, 0, st>>>(s); })
    }
    HAN_CHECK_LAUNCH();
    return 0;
}

// multi-meta-path eval forward on the fused kernel: groups of 4, then 2 meta-paths per block
template <bool XBF>
static int launch_multi(ProjMultiArgs m, int P, int np_max, hipStream_t st, int *done) {
    *done = 0;      // returns 0 or an error code (HIP errors are small POSITIVE ints: never to be read as a count)
    constexpr int MT = 2;
    const unsigned tiles = (unsigned)((m.N + 128 * MT - 1) / (128 * MT));
    const size_t xb = (size_t)(XBF ? 1 : 3) * 128 * MT * B6_LDB;
    int p = 0;
    while (P - p >= 2) {
        const int np = (P - p >= 4 && np_max >= 4) ? 4 : 2;
        const int groups = (P - p) / np;
        const size_t lds = xb + (size_t)3 * np * B6_WBYTES;
        m.p_first = p;
        hipError_t e;
        if (np == 4) {
            e = hipFuncSetAttribute((const void *)project_fwd_b6_multi_kernel<XBF, 4, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            project_fwd_b6_multi_kernel<XBF, 4, MT><<<dim3(tiles, groups), 512, lds, st>>>(m);
        } else {
            e = hipFuncSetAttribute((const void *)project_fwd_b6_multi_kernel<XBF, 2, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            project_fwd_b6_multi_kernel<XBF, 2, MT><<<dim3(tiles, groups), 512, lds, st>>>(m);
        }
        HAN_CHECK_LAUNCH();
        p += groups * np;
        *done = p;      // meta-paths done; a last odd one is left to the caller
    }
    return 0;
}

extern "C" int han_project_fwd_multi(const void *X, int x_dtype, int64_t ldx, const float *W, const float *a1,
                                     const float *a2, const float *b1, const float *b2, void *H, int table_dtype,
                                     float *f1, float *f2, void *workspace, size_t workspace_bytes, int64_t N,
                                     int F, int K, int FP, int P, float in_drop, float fts_drop,
                                     const uint64_t *seeds, const uint64_t *seed_dev, int64_t row_offset,
                                     uint8_t *keep, int flags, void *stream) {
    if (P <= 0 || (in_drop > 0.f && !seeds)) return HAN_E_BADARG;
    if (N == 0) return 0;
    if (!X || !W || !a1 || !a2 || !b1 || !b2 || !H || !f1 || !f2 || N < 0 || F <= 0 || ldx < F) return HAN_E_BADARG;
    if (!fp_ok(K, FP)) return HAN_E_UNSUPPORTED;
    const bool x_bf16 = x_dtype == HAN_DTYPE_BF16, h_bf16 = table_dtype == HAN_DTYPE_BF16;
    const size_t hb = (size_t)N * HAN_D * (h_bf16 ? 2 : 4), kb = han_project_keep_bytes(N, F, ldx, K, FP);
    int mt, nsplit, fch;
    fwd_geometry(N, F, &mt, &nsplit, &fch);
    const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)X & (x_bf16 ? 7 : 15)) == 0);
    int done = 0;
    if (in_drop == 0.f && fts_drop == 0.f && P >= 2 && vec && nsplit == 1 && N >= 64 * 256 &&
        !(flags & HAN_FLAG_K1_EXACT_PIPE) && (x_dtype == HAN_DTYPE_F32 || x_dtype == HAN_DTYPE_BF16) &&
        (table_dtype == HAN_DTYPE_F32 || h_bf16)) {
        hipStream_t st = (hipStream_t)stream;
        if (!workspace || workspace_bytes < wimage_bytes(F, P)) return HAN_E_WORKSPACE;
        const int ktiles = (F + 31) / 32;
        project_wimage_kernel<<<(int)(((int64_t)ktiles * 512 * P + 255) / 256), 256, 0, st>>>(W, (unsigned char *)workspace, F, ktiles, P);
        HAN_CHECK_LAUNCH();
        ProjMultiArgs m;
        m.wimg = (const unsigned char *)workspace; m.P = P;
        m.X = X; m.ldx = ldx; m.W = W; m.H = H; m.h_bf16 = h_bf16; m.N = N; m.F = F; m.p_first = 0;
        m.a1 = a1; m.a2 = a2; m.b1 = b1; m.b2 = b2; m.f1 = f1; m.f2 = f2; m.K = K; m.fuse_scores = FP == 8;
        const int np_max = (flags & HAN_FLAG_K1_PAIRS) ? 2 : 4;
        const int rc = x_bf16 ? launch_multi<true>(m, P, np_max, st, &done) : launch_multi<false>(m, P, np_max, st, &done);
        if (rc != 0) return rc;
        if (FP != 8) {                               // other head widths: scores from the stored rows
            for (int p = 0; p < done; ++p) {
                ScoreArgs s;
                s.H = (const char *)H + (size_t)p * hb; s.a1 = a1 + (size_t)p * HAN_D; s.a2 = a2 + (size_t)p * HAN_D;
                s.b1 = b1 + (size_t)p * K; s.b2 = b2 + (size_t)p * K;
                s.f1 = f1 + (size_t)p * N * K; s.f2 = f2 + (size_t)p * N * K; s.N = N;
                const int sgrid = han_grid_for(N, 16, 256 * 8);
                if (h_bf16) { HAN_DISPATCH_FP(FP, { project_scores_kernel<FPC, true><<<sgrid, 256, 0, st>>>(s); }) }
                else { HAN_DISPATCH_FP(FP, { project_scores_kernel<FPC, false><<<sgrid, 256, 0, st>>>(s); }) }
                HAN_CHECK_LAUNCH();
            }
        }
    }
    for (int p = done; p < P; ++p) {                 // everything else: one meta-path at a time
        const int rc = han_project_fwd(X, x_dtype, ldx, W + (size_t)p * F * HAN_D, a1 + (size_t)p * HAN_D,
                                       a2 + (size_t)p * HAN_D, b1 + (size_t)p * K, b2 + (size_t)p * K,
                                       (char *)H + (size_t)p * hb, table_dtype, f1 + (size_t)p * N * K,
                                       f2 + (size_t)p * N * K, workspace, workspace_bytes, N, F, K, FP, in_drop,
                                       fts_drop, seeds ? seeds[p] : 0, seed_dev, row_offset,
                                       (keep && kb) ? keep + (size_t)p * kb : nullptr, flags, stream);
        if (rc != 0) return rc;
    }
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
                               uint64_t seed, const uint64_t *seed_dev, int64_t row_offset, const uint8_t *keep,
                               void *stream) {
    if (!X || !dH || !dW || !workspace || N < 0 || F <= 0 || ldx < F) return HAN_E_BADARG;
    if (!fp_ok(K, FP)) return HAN_E_UNSUPPORTED;
    if (x_dtype != HAN_DTYPE_F32 && x_dtype != HAN_DTYPE_BF16) return HAN_E_UNSUPPORTED;
    if (in_drop < 0.f || in_drop >= 1.f) return HAN_E_BADARG;
    if (workspace_bytes < han_project_bwd_workspace(N, F, K, FP)) return HAN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int ftiles; int64_t rpc, nch;
    bwd_geometry(N, F, &ftiles, &rpc, &nch);
    const bool x_bf16 = x_dtype == HAN_DTYPE_BF16;
    // 16-byte fp32 / 8-byte bf16 X loads (the scalar path costs 1.7x at SYN-10M: 14.3 vs 8.4 ms per launch)
    const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)X & (x_bf16 ? 7 : 15)) == 0);
    if (keep && in_drop > 0.f && !(vec && keep_table_shape(N, F, ldx, K, FP) && ((uintptr_t)keep & 7) == 0))
        return HAN_E_BADARG;      // a table exists only for the shapes han_project_keep_bytes() names
    const int width = F * HAN_D;
    if (keep && in_drop > 0.f && N > 0) {
        // the 16-block 4x4x1 kernel: the forward's keep words are the A operand's lane masks (no draw is regenerated)
        ProjBwdBlkArgs b;
        b.X = X; b.ldx = ldx; b.dH = dH; b.keep = keep; b.slab = (float *)workspace; b.N = N; b.F = F;
        b.rows_per_chunk = rpc; b.inv_keep_in = 1.f / (1.f - in_drop);
        const dim3 gb((unsigned)((F + DW_FB - 1) / DW_FB), (unsigned)nch);
        if (x_bf16) project_bwd_blk_kernel<true, true><<<gb, 256, 0, st>>>(b);
        else project_bwd_blk_kernel<true, false><<<gb, 256, 0, st>>>(b);
        HAN_CHECK_LAUNCH();
        hipError_t e = han_reduce_slabs((const float *)workspace, (int)nch, width, width, han_reduce_to(dW, width), st);
        return e != hipSuccess ? (int)e : 0;
    }
    ProjBwdArgs a;
    a.X = X; a.x_bf16 = x_bf16; a.ldx = ldx; a.dH = dH; a.slab = (float *)workspace; a.N = N; a.F = F;
    a.rows_per_chunk = rpc;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.seed_dev = seed_dev;
    a.thr_in = in_drop > 0.f ? han_keep_threshold(1.f - in_drop) : HAN_KEEP_ALL;
    a.inv_keep_in = 1.f / (1.f - in_drop);
    a.row_offset = row_offset;
    dim3 grid(ftiles, (unsigned)nch);
    // (dW on the bf16 x 6 matrix pipe was built and measured in round 2 -- coalesced loads + on-chip transpose:
    // 0.55 ms without / 0.83 ms with dropout against 0.41 / 0.84 ms for this exact-fp32 kernel at SYN-1M --
    // and not kept: the transposition of both operands through LDS costs what the shorter matrix time saves.)
    const bool vec32 = true;      // quad loads at element alignment (load_x4_tail)
    HAN_DISPATCH_FP(FP, {
        if (in_drop > 0.f) {
            if (vec32) project_bwd_kernel<FPC, true, kBwdMT, true><<<grid, 256, 0, st>>>(a);
            else project_bwd_kernel<FPC, true, kBwdMT, false><<<grid, 256, 0, st>>>(a);
        } else {
            if (vec32) project_bwd_kernel<FPC, false, kBwdMT, true><<<grid, 256, 0, st>>>(a);
            else project_bwd_kernel<FPC, false, kBwdMT, false><<<grid, 256, 0, st>>>(a);
        }
    })
    HAN_CHECK_LAUNCH();
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
