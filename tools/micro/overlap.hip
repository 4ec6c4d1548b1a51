// How much vector-ALU work hides under MFMAs on gfx950?  (Round 4: K1's bf16 x 6 kernels and the dense K2 kernels both
// run at "matrix time + vector time", not max(...) -- profiles/r04_k2_dense_experiments.md.)
// A wave issues, per MFMA, KV independent v_fma_f32 (inline asm, fixed order) and one MFMA on its own accumulator
// (8 accumulators round-robin: no dependent-issue stall); s_memtime ticks per MFMA slot and their ratio to the same
// configuration without vector instructions.  Cases: MFMA type (f32 16x16x4: 32 cycles, bf16 16x16x32: 16),
// KV = 0 .. 12, 1 / 2 / 4 waves per SIMD, and "split": half of the waves issue ONLY the MFMAs, the other half ONLY the
// vector instructions (what a producer / consumer specialisation would look like).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/overlap tools/micro/overlap.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define ITERS 128

template <int MF, int KV, int ROLE>      // ROLE 0: both in every wave; 1: even waves MFMA only, odd waves VALU only
__global__ __launch_bounds__(256) void overlap_kernel(uint64_t *out, float seed) {
    f32x4 acc[8];
    float v[12];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
    for (int i = 0; i < 12; ++i) v[i] = seed + i + threadIdx.x;
    const float a = seed * 0.25f, b = seed * 0.5f;
    i32x4 ab = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = ROLE == 0 || (wave & 1) == 0;
    const bool do_valu = ROLE == 0 || (wave & 1) == 1;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (do_valu) {
#pragma unroll
                for (int k = 0; k < KV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            }
            if (do_mfma) {
                if (MF == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(ab));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    for (int i = 0; i < 12; ++i) s += v[i];
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
    if (s == 0.123f) out[0] = 1;
}

static double mean(const uint64_t *h, int n, int stride = 1, int first = 0) {
    double s = 0; int c = 0;
    for (int i = first; i < n; i += stride) { s += (double)h[i]; ++c; }
    return s / c;
}

int main() {
    uint64_t *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    uint64_t *h = (uint64_t *)malloc(1 << 20);
    const int nb = 256;      // 256 blocks of 4 waves: one wave per SIMD; x2 / x4 for more
    static double base[2][2][5];      // ticks of the KV = 0 run of the same (mfma, role, waves) configuration
    // s_memtime ticks are shader cycles at whatever clock the run sustains: report ticks and the ratio to the MFMA-only
    // run, not "cycles" (one wave per SIMD: 32.8 / 17.0 ticks per f32 16x16x4 / bf16 16x16x32 MFMA alone)
#define RUN(MF, KV, ROLE, MULT)                                                                                          \
    {                                                                                                                    \
        overlap_kernel<MF, KV, ROLE><<<nb * MULT, 256>>>(d, 1.0f); hipDeviceSynchronize();                                \
        overlap_kernel<MF, KV, ROLE><<<nb * MULT, 256>>>(d, 1.0f); hipDeviceSynchronize();                                \
        hipMemcpy(h, d, nb * MULT * 4 * 8, hipMemcpyDeviceToHost);                                                        \
        const double ticks = mean(h, nb * MULT * 4) / (ITERS * 8.0);                                                      \
        if (KV == 0) base[MF][ROLE][MULT] = ticks;                                                                        \
        printf("{\"mfma\": \"%s\", \"issue\": \"%s\", \"waves_per_simd\": %d, \"valu_per_mfma\": %d, "                   \
               "\"ticks_per_mfma_slot_per_wave\": %.1f, \"ratio_to_mfma_only\": %.2f}\n",                                  \
               MF ? "bf16 16x16x32" : "f32 16x16x4", ROLE ? "split waves" : "same wave", MULT, KV, ticks,                  \
               ticks / base[MF][ROLE][MULT]);                                                                             \
        fflush(stdout);                                                                                                   \
    }
#define SWEEP(MF, ROLE, MULT) RUN(MF, 0, ROLE, MULT) RUN(MF, 2, ROLE, MULT) RUN(MF, 4, ROLE, MULT) RUN(MF, 6, ROLE, MULT) \
    RUN(MF, 8, ROLE, MULT) RUN(MF, 12, ROLE, MULT)
    SWEEP(0, 0, 1) SWEEP(0, 0, 2) SWEEP(0, 0, 4)
    SWEEP(1, 0, 1) SWEEP(1, 0, 2) SWEEP(1, 0, 4)
    SWEEP(0, 1, 2) SWEEP(0, 1, 4) SWEEP(1, 1, 2) SWEEP(1, 1, 4)
    return 0;
}
