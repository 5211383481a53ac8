// Does the two-wide single-precision fused multiply-add of gfx950 (v_pk_fma_f32) issue in the slot of a scalar one
// (v_fma_f32), or does it take two?  Both kernels run the same number of dependent-free multiply-adds per lane, the
// first as N scalar instructions, the second as N / 2 packed ones; eight waves per SIMD hide the latency, so time is
// issue slots.
//
//   hipcc --offload-arch=gfx950 -O3 -o pk_fma_probe pk_fma_probe.hip && ./pk_fma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float pair_t __attribute__((ext_vector_type(2)));

#define CHECK(e)                                                    \
    do {                                                            \
        hipError_t e_ = (e);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); \
            return 1;                                               \
        }                                                           \
    } while (0)

constexpr int ITERS = 65536;

__global__ __launch_bounds__(256) void scalar_fma(float *out, float m, float c) {
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (float)(threadIdx.x + k);
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c)); // the compiler would pair them
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(256) void packed_fma(float *out, float m, float c) {
    pair_t a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = pair_t{(float)(threadIdx.x + k), (float)(threadIdx.x + k + 4)};
    const pair_t mm = {m, m}, cc = {c, c};
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = __builtin_elementwise_fma(a[k], mm, cc);
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += a[k].x + a[k].y;
    if (s == 123.456f) out[0] = s;
}

// Issue cost of single instructions: one wave alone on its SIMD, eight independent chains, core clock cycles by
// s_memtime.  cycles / instruction = how long the SIMD is occupied by one wave64 instruction of that kind.
#define ONE_WAVE_KERNEL(NAME, ASM)                                                                    \
    __global__ __launch_bounds__(64) void NAME(unsigned long long *cycles, float m, float c) {        \
        float a[8];                                                                                   \
        for (int k = 0; k < 8; ++k) a[k] = (float)(threadIdx.x + k);                                  \
        const unsigned long long t0 = __builtin_readcyclecounter();                                   \
        for (int i = 0; i < 1024; ++i) {                                                              \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(m), "v"(c)); \
        }                                                                                             \
        const unsigned long long t1 = __builtin_readcyclecounter();                                   \
        float s = 0.0f;                                                                               \
        for (int k = 0; k < 8; ++k) s += a[k];                                                        \
        if (threadIdx.x == 0) cycles[0] = t1 - t0;                                                    \
        if (s == 123.456f) cycles[1] = 1;                                                             \
    }
ONE_WAVE_KERNEL(one_fma, "v_fma_f32 %0, %0, %1, %2")
ONE_WAVE_KERNEL(one_mul, "v_mul_f32 %0, %0, %1")
ONE_WAVE_KERNEL(one_max3, "v_max3_f32 %0, %0, %1, %2")
ONE_WAVE_KERNEL(one_cvt, "v_cvt_f32_ubyte1 %0, %0")
ONE_WAVE_KERNEL(one_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
ONE_WAVE_KERNEL(one_cmp, "v_cmp_lt_f32 vcc, %0, %1")

__global__ __launch_bounds__(64) void one_pk_fma(unsigned long long *cycles, float m, float c) {
    pair_t a[8];
    for (int k = 0; k < 8; ++k) a[k] = pair_t{(float)(threadIdx.x + k), (float)(threadIdx.x + k + 4)};
    const pair_t mm = {m, m}, cc = {c, c};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 1024; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(mm), "v"(cc));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += a[k].x + a[k].y;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
    if (s == 123.456f) cycles[1] = 1;
}

template <class K>
static int one_wave(const char *name, K kernel) {
    unsigned long long *d, h = 0;
    CHECK(hipMalloc(&d, 64));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, 0, d, 0.999f, 0.001f);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("one wave, %-14s %.2f cycles per instruction\n", name, (double)h / (1024.0 * 8.0));
    CHECK(hipFree(d));
    return 0;
}

int main() {
    if (one_wave("v_fma_f32", one_fma) || one_wave("v_pk_fma_f32", one_pk_fma) || one_wave("v_mul_f32", one_mul) ||
        one_wave("v_max3_f32", one_max3) || one_wave("v_cvt_f32_ubyte", one_cvt) || one_wave("v_cndmask_b32", one_cndmask) ||
        one_wave("v_cmp_lt_f32", one_cmp))
        return 1;
    float *out;
    CHECK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 8; // eight waves per SIMD
    for (int which = 0; which < 2; ++which) {
        float ms = 0.0f;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            if (which == 0)
                hipLaunchKernelGGL(scalar_fma, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
            else
                hipLaunchKernelGGL(packed_fma, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double fmas = (double)blocks * 256.0 * ITERS * 8.0;
        printf("%s  %.3f ms  %.1f G multiply-adds/s per lane-op  (%.2f T multiply-adds/s)\n", which == 0 ? "v_fma_f32   " : "v_pk_fma_f32",
               ms, fmas / ms / 1e6, fmas / ms / 1e9);
    }
    return 0;
}
