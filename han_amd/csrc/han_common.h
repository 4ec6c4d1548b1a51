// Internal helpers shared by the gfx950 kernels of libhan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/han_hip.h"

#define HAN_D 64            // K*FP of this build: one projected row = 256 B
#define HAN_NEG_BIG (-1.0e30f)

#define HAN_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t _e = hipGetLastError();                   \
        if (_e != hipSuccess) return (int)_e;                \
    } while (0)

typedef float float4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Counter-based dropout RNG: two rounds of the 32x32->64 multiply-fold ("mum", as in
// wyhash32) keyed by (seed, stream) over the counter (a, b).  One call costs four
// integer multiplies and yields 64 bits = FOUR 16-bit Bernoulli draws (keep iff
// field < keep_prob * 2^16); forward, backward and any node partition regenerate the
// same mask from global ids.  tests/rng_ref.py restates it in NumPy so that the
// oracle can be fed the identical masks.  (Measured alternatives: a murmur-style
// 32-bit hash with two draws per call, and an add/rotate/xor Threefry-2x32 -- both
// slower in the K1/K2 training kernels.)
//   stream 0  input-feature dropout (layers.py:19)   a = global row, b = f*ceil(K/4) + k/4,  field k%4
//   stream 1  attention dropout     (layers.py:30)   a = global dst i, b = j*ceil(K/4) + k/4, field k%4
//   stream 2  projected-row dropout (layers.py:32)   a = global row j, b = d/4,               field d%4
// ---------------------------------------------------------------------------
#define HAN_STREAM_SEQ 0u
#define HAN_STREAM_COEF 1u
#define HAN_STREAM_FTS 2u

struct HanRand64 {
    uint32_t x, y;
    // 16-bit field f in [0,4)
    __host__ __device__ __forceinline__ uint32_t field(int f) const {
        return (((f & 2) ? y : x) >> (16 * (f & 1))) & 0xFFFFu;
    }
};

__host__ __device__ __forceinline__ void han_mum(uint32_t &A, uint32_t &B) {
    const uint64_t c = (uint64_t)(A ^ 0x53C5CA59u) * (uint64_t)(B ^ 0x74743C1Bu);
    A = (uint32_t)c;
    B = (uint32_t)(c >> 32);
}

__host__ __device__ __forceinline__ HanRand64 han_rand64(uint32_t seed_lo, uint32_t seed_hi, uint32_t stream,
                                                         uint32_t a, uint32_t b) {
    uint32_t A = a ^ seed_lo, B = b ^ (seed_hi + stream * 0x9E3779B9u);
    han_mum(A, B);
    A ^= seed_hi;
    B ^= seed_lo;
    han_mum(A, B);
    HanRand64 r;
    r.x = A ^ B;          // fold the halves once more so every output bit sees both
    r.y = B ^ (A >> 15) ^ (A << 17);
    return r;
}

// A captured hipGraph replays fixed kernel arguments, so a by-value seed would repeat the
// same masks every step.  Every seeded entry point therefore also takes `seed_dev`: when
// non-null the effective seed is splitmix64(seed + *seed_dev), read on the device at
// kernel start -- the host bumps the device word between replays (a captured add).
__host__ __device__ __forceinline__ uint64_t han_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ void han_resolve_seed(uint32_t &seed_lo, uint32_t &seed_hi, const uint64_t *seed_dev) {
    if (seed_dev) {
        const uint64_t s = han_splitmix64((((uint64_t)seed_hi << 32) | seed_lo) + *seed_dev);
        seed_lo = (uint32_t)s;
        seed_hi = (uint32_t)(s >> 32);
    }
}

// keep iff the 16-bit field is below keep_prob * 2^16 (65536 = keep everything)
__host__ __device__ __forceinline__ uint32_t han_keep_threshold(float keep_prob) {
    return (uint32_t)(keep_prob * 65536.0f);
}
#define HAN_KEEP_ALL 65536u

// ---------------------------------------------------------------------------
// Table rows in fp32 (256 B) or bf16 (128 B): lane q of a 16-lane group owns the 4
// features 4q..4q+3 -> one 16-B or 8-B load.  bf16 is widened by a 16-bit shift
// (exact), rounded to nearest-even on store.  The dropout keep bit is the LOWEST
// mantissa bit of the stored element, i.e. bit 0 (fp32) or bit 16 of the widened
// float (bf16).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t han_f32_to_bf16_bits(float f) {
    const uint32_t u = __float_as_uint(f);
    // a NaN stays a (quiet) NaN: the rounding add would carry 0x7FFFxxxx into the sign bit
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

template <bool BF>
__device__ __forceinline__ float4_t han_load_row4(const void *tab, int64_t row, int q) {
    if (BF) {
        const uint2 w = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(tab) + row * 64 + 4 * q);
        float4_t v;
        v[0] = __uint_as_float(w.x << 16);
        v[1] = __uint_as_float(w.x & 0xFFFF0000u);
        v[2] = __uint_as_float(w.y << 16);
        v[3] = __uint_as_float(w.y & 0xFFFF0000u);
        return v;
    }
    return *reinterpret_cast<const float4_t *>(reinterpret_cast<const float *>(tab) + row * 64 + 4 * q);
}

template <bool BF>
__device__ __forceinline__ void han_store_row4(void *tab, int64_t row, int q, const float4_t &v) {
    if (BF) {
        uint2 w;
        w.x = han_f32_to_bf16_bits(v[0]) | (han_f32_to_bf16_bits(v[1]) << 16);
        w.y = han_f32_to_bf16_bits(v[2]) | (han_f32_to_bf16_bits(v[3]) << 16);
        *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(tab) + row * 64 + 4 * q) = w;
    } else {
        *reinterpret_cast<float4_t *>(reinterpret_cast<float *>(tab) + row * 64 + 4 * q) = v;
    }
}

template <bool BF>
__device__ __forceinline__ uint32_t han_keep_bit(float v) {
    return (__float_as_uint(v) >> (BF ? 16 : 0)) & 1u;
}

__device__ __forceinline__ float han_lrelu(float x, float slope) { return fmaxf(x, slope * x); }

__device__ __forceinline__ float han_elu(float x) { return x > 0.f ? x : (__expf(x) - 1.f); }

template <typename T>
__device__ __forceinline__ T han_shfl_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }

// sum over an aligned group of 16 lanes, every lane gets the total: two quad permutes and
// two row rotations, all DPP modifiers of a v_add_f32 (no ds_bpermute, no LDS traffic)
__device__ __forceinline__ float han_row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));  // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));  // row_ror:8
    return v;
}

// v of lane (i ^ O) inside an aligned group of 16 lanes, O = 1, 2, 4 or 8, as a DPP operand (no ds_bpermute, no LDS
// round trip).  O = 4 is row_half_mirror (lane 7 - i of the 8-lane half): the same VALUE as lane i ^ 4 whenever the four
// lanes of a quad hold equal values -- which they do after the O = 1 and O = 2 steps of a butterfly sum, the only use.
template <int O>
__device__ __forceinline__ float han_dpp_xor16(float v) {
    static_assert(O == 1 || O == 2 || O == 4 || O == 8, "lane distance inside a row of 16");
    constexpr int ctrl = O == 1 ? 0xB1 : O == 2 ? 0x4E : O == 4 ? 0x141 : 0x128;   // quad_perm x2, row_half_mirror, row_ror:8
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true));
}

__device__ __forceinline__ float han_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------
// Deterministic second stage of every cross-block reduction:
//   out[n] = scale * sum_b slab[b*row_stride + n],  n in [0, width)
// routed to up to 4 output segments (segment i covers [seg_end[i-1], seg_end[i]))
// and replicated `rep` times at stride rep_stride (classifier heads).  A block
// owns 16 consecutive n; its 16 slices of blocks are summed in a fixed order.
// ---------------------------------------------------------------------------
struct HanReduceOut {
    float *ptr[4];
    int seg_end[4];
    int nseg;
    float scale[4];          // per segment
    int rep[4];              // per segment: replicas written at stride rep_stride[seg]
    int64_t rep_stride[4];
};

static __global__ __launch_bounds__(256) void han_reduce_slabs_kernel(const float *slab, int nblocks,
                                                                     int64_t row_stride, int width,
                                                                     HanReduceOut o) {
    __shared__ float part[16][17];
    const int nn = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + nn;
    // four independent partial sums per thread (rows sl, sl + 16, sl + 32, sl + 48 (mod 64)): the loads of a step do
    // not wait for each other -- at the slab counts of small graphs the chain of dependent L2 round trips WAS the
    // kernel (8 us for a 64-float result); the order of the additions is fixed, so the result stays reproducible
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < width) {
        const float *col = slab + n;
        int b = sl;
        // (round 4) two steps = eight loads in flight before the first add: at the slab counts of the small graphs
        // (100-300 rows) the loop used to be 2-4 dependent L2 round trips; the additions keep their order
        for (; b + 112 < nblocks; b += 128) {
            const float v0 = col[(int64_t)b * row_stride], v1 = col[(int64_t)(b + 16) * row_stride];
            const float v2 = col[(int64_t)(b + 32) * row_stride], v3 = col[(int64_t)(b + 48) * row_stride];
            const float v4 = col[(int64_t)(b + 64) * row_stride], v5 = col[(int64_t)(b + 80) * row_stride];
            const float v6 = col[(int64_t)(b + 96) * row_stride], v7 = col[(int64_t)(b + 112) * row_stride];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
            s0 += v4; s1 += v5; s2 += v6; s3 += v7;
        }
        for (; b + 48 < nblocks; b += 64) {
            s0 += col[(int64_t)b * row_stride];
            s1 += col[(int64_t)(b + 16) * row_stride];
            s2 += col[(int64_t)(b + 32) * row_stride];
            s3 += col[(int64_t)(b + 48) * row_stride];
        }
        for (; b < nblocks; b += 16) s0 += col[(int64_t)b * row_stride];
    }
    part[sl][nn] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && n < width) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += part[r][nn];
        int seg = 0, base = 0;
        while (seg < o.nseg - 1 && n >= o.seg_end[seg]) { base = o.seg_end[seg]; ++seg; }
        t *= o.scale[seg];
        for (int r = 0; r < o.rep[seg]; ++r) o.ptr[seg][(int64_t)r * o.rep_stride[seg] + (n - base)] = t;
    }
}

// slab rows are `row_stride` floats apart; columns [0, width) of each row are reduced
static inline hipError_t han_reduce_slabs(const float *slab, int nblocks, int64_t row_stride, int width,
                                          const HanReduceOut &o, hipStream_t st) {
    han_reduce_slabs_kernel<<<(width + 15) / 16, 256, 0, st>>>(slab, nblocks, row_stride, width, o);
    return hipGetLastError();
}

static inline HanReduceOut han_reduce_to(float *out, int width) {
    HanReduceOut o;
    o.ptr[0] = out; o.ptr[1] = o.ptr[2] = o.ptr[3] = nullptr;
    o.seg_end[0] = width; o.seg_end[1] = o.seg_end[2] = o.seg_end[3] = width;
    o.nseg = 1;
    for (int i = 0; i < 4; ++i) { o.scale[i] = 1.f; o.rep[i] = 1; o.rep_stride[i] = 0; }
    return o;
}

static inline int han_grid_for(int64_t items, int per_block, int cap) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}
