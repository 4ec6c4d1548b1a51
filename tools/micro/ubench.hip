// Micro-measurements used while designing the round-3 K1 kernels (gfx950):
//   1. issue cost of the integer multiplies the dropout hash is built from
//   2. operand / result lane maps and issue cost of v_mfma_f32_4x4x1_16B_f32
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/ubench tools/micro/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP 64
template <int WHICH>
__global__ __launch_bounds__(256) void issue_kernel(uint64_t *out, uint32_t seed) {
    uint32_t a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i * 11 + threadIdx.x; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (WHICH == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 4) {
                    uint64_t d;
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a[i]), "v"(b[i]) : "vcc");
                    a[i] = (uint32_t)d; b[i] ^= (uint32_t)(d >> 32);
                }
                if (WHICH == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 6) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 7) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 8) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 9) asm volatile("v_bfe_i32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 11) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 12) asm volatile("v_alignbit_b32 %0, %0, %1, 15" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 13) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 14) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (WHICH == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i] ^ b[i];
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t1 - t0;
    if (s == 0x12345678u) out[1] = s;
}

__global__ void mfma_map_kernel(float *d) {
    const int l = threadIdx.x;
    const float a = 65.f + l, b = exp2f(-(float)l);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate_kernel(uint64_t *out, float seed) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t1 - t0;
    if (s == 0.123f) out[1] = 1;
}

// the same with one v_cndmask (SGPR-pair lane mask) per MFMA feeding the A operand
template <int NACC>
__global__ __launch_bounds__(256) void mfma_mask_rate_kernel(uint64_t *out, float seed, const uint64_t *masks) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it) {
        const uint64_t r0 = masks[(it * 2) & 63], r1 = masks[(it * 2 + 1) & 63];
        const uint64_t m0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(r0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)r0);
        const uint64_t m1 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(r1 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)r1);
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            float am;
            asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(am) : "v"(a), "s"((i & 1) ? m1 : m0));
            acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(am, b, acc[i], 0, 0, 0);
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t1 - t0;
    if (s == 0.123f) out[1] = 1;
}

static double med(uint64_t *h, int n) {
    double s = 0; int c = 0;
    for (int i = 0; i < n; ++i) { s += (double)h[2 * i]; ++c; }
    return s / c;
}

int main() {
    uint64_t *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    uint64_t *h = (uint64_t *)malloc(1 << 20);
    const char *names[] = {"v_xor_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u64_u32(+xor)",
                           "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_pk_sub_u16 clamp", "v_perm_b32", "v_bfe_i32",
                           "v_and_b32", "v_pk_min_u16", "v_alignbit_b32", "v_sub_f32", "v_and_or_b32", "v_cndmask_b32"};
    const int nb = 256;   // one 4-wave block per CU -> one wave per SIMD
#define RUN(W) { issue_kernel<W><<<nb, 256>>>(d, 12345u); hipDeviceSynchronize(); issue_kernel<W><<<nb, 256>>>(d, 12345u); \
        hipDeviceSynchronize(); hipMemcpy(h, d, nb * 4 * 16, hipMemcpyDeviceToHost); \
        printf("{\"instr\": \"%s\", \"memtime_ticks_per_instr_one_wave_per_simd\": %.2f}\n", names[W], med(h, nb * 4) / (64.0 * REP)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
    // two waves per SIMD (two blocks per CU)
#define RUN2(W) { issue_kernel<W><<<nb * 2, 256>>>(d, 12345u); hipDeviceSynchronize(); issue_kernel<W><<<nb * 2, 256>>>(d, 12345u); \
        hipDeviceSynchronize(); hipMemcpy(h, d, nb * 2 * 4 * 16, hipMemcpyDeviceToHost); \
        printf("{\"instr\": \"%s\", \"memtime_ticks_per_instr_two_waves_per_simd\": %.2f}\n", names[W], med(h, nb * 8) / (64.0 * REP)); }
    RUN2(0) RUN2(1) RUN2(4) RUN2(5)
    // four waves per SIMD: the SIMD's throughput per instruction = ticks / 4
#define RUN4(W) { issue_kernel<W><<<nb * 4, 256>>>(d, 12345u); hipDeviceSynchronize(); issue_kernel<W><<<nb * 4, 256>>>(d, 12345u); \
        hipDeviceSynchronize(); hipMemcpy(h, d, nb * 4 * 4 * 16, hipMemcpyDeviceToHost); \
        printf("{\"instr\": \"%s\", \"memtime_ticks_per_instr_four_waves_per_simd\": %.2f, \"simd_cycles_per_instr\": %.2f}\n", names[W], med(h, nb * 16) / (64.0 * REP), med(h, nb * 16) / (64.0 * REP) / 4); }
    RUN4(0) RUN4(1) RUN4(4) RUN4(7) RUN4(8) RUN4(9) RUN4(10) RUN4(11) RUN4(12) RUN4(13) RUN4(14) RUN4(15)
    float *dm; hipMalloc(&dm, 256 * 4);
    mfma_map_kernel<<<1, 64>>>(dm); hipDeviceSynchronize();
    float hm[256]; hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost);
    printf("4x4x1_16B map: lane reg -> (A lane, B lane)\n");
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            int e; float m = frexpf(hm[l * 4 + r], &e);   // v = (65+x) * 2^-y
            // (65+x) in [65,128]: m*128 = 65+x when m in (0.5,1); then e = 7 - y
            int x, y;
            if (m == 0.5f) { x = 63; y = 8 - e; } else { x = (int)lrintf(m * 128.f) - 65; y = 7 - e; }
            printf("  r%d=(A%2d,B%2d)", r, x, y);
        }
        printf("\n");
    }
    uint64_t *dmask; hipMalloc(&dmask, 64 * 8);
    uint64_t hmask[64]; for (int i = 0; i < 64; ++i) hmask[i] = 0x9E3779B97F4A7C15ull * (i + 1);
    hipMemcpy(dmask, hmask, sizeof(hmask), hipMemcpyHostToDevice);
#define RUNM(K, N, ...) { K<N><<<nb, 256>>>(__VA_ARGS__); hipDeviceSynchronize(); K<N><<<nb, 256>>>(__VA_ARGS__); hipDeviceSynchronize(); \
        hipMemcpy(h, d, nb * 4 * 16, hipMemcpyDeviceToHost); \
        printf("{\"kernel\": \"%s<%d>\", \"memtime_ticks_per_mfma\": %.2f}\n", #K, N, med(h, nb * 4) / (256.0 * N)); }
    RUNM(mfma_rate_kernel, 1, d, 1.0f) RUNM(mfma_rate_kernel, 4, d, 1.0f) RUNM(mfma_rate_kernel, 8, d, 1.0f)
    RUNM(mfma_mask_rate_kernel, 8, d, 1.0f, dmask)
    printf("note: s_memtime ticks at 100 MHz on gfx950? compare v_xor (4 shader cycles expected for one wave)\n");
    return 0;
}
