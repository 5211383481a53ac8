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

// The same streams on the whole chip, eight waves per SIMD: wave instructions per second of each kind.
#define FULL_KERNEL(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, float m, float c) {                       \
        float a[8];                                                                                   \
        for (int k = 0; k < 8; ++k) a[k] = (float)(threadIdx.x + k);                                  \
        for (int i = 0; i < 16384; ++i) {                                                             \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(m), "v"(c)); \
        }                                                                                             \
        float s = 0.0f;                                                                               \
        for (int k = 0; k < 8; ++k) s += a[k];                                                        \
        if (s == 123.456f) out[0] = s;                                                                \
    }
FULL_KERNEL(full_fma, "v_fma_f32 %0, %0, %1, %2")
FULL_KERNEL(full_mul, "v_mul_f32 %0, %0, %1")
FULL_KERNEL(full_min, "v_min_f32 %0, %0, %1")
FULL_KERNEL(full_max3, "v_max3_f32 %0, %0, %1, %2")
FULL_KERNEL(full_cvt, "v_cvt_f32_ubyte1 %0, %0")
FULL_KERNEL(full_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
FULL_KERNEL(full_cmp, "v_cmp_lt_f32 vcc, %0, %1")
FULL_KERNEL(full_addu, "v_add_u32 %0, %0, %1")
FULL_KERNEL(full_mov, "v_mov_b32 %0, %1")
FULL_KERNEL(full_minu, "v_min_u32 %0, %0, %1")
FULL_KERNEL(full_and, "v_and_b32 %0, %0, %1")
FULL_KERNEL(full_bfe, "v_bfe_u32 %0, %0, 8, 8")
FULL_KERNEL(full_perm, "v_perm_b32 %0, %0, %1, %2")
FULL_KERNEL(full_med3, "v_med3_f32 %0, %0, %1, %2")
FULL_KERNEL(full_cndmask_s, "v_cndmask_b32 %0, %0, %1, s[10:11]")

// v_cndmask_b32 with a condition that was computed (vcc written by a compare shortly before, as in real code)
__global__ __launch_bounds__(256) void full_cndmask_fresh(float *out, float m, float c) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = (float)(threadIdx.x + k);
    for (int i = 0; i < 16384; ++i) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(m) : "vcc");
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c) : "vcc");
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 123.456f) out[0] = s;
}

// compare + select pairs, the condition in vcc (what the compiler emits for `c ? a : b`) or in a scalar register pair
__global__ __launch_bounds__(256) void full_select_vcc(float *out, float m, float c) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = (float)(threadIdx.x + k);
    for (int i = 0; i < 16384; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(m), "v"(c) : "vcc");
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void full_select_sgpr(float *out, float m, float c) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = (float)(threadIdx.x + k);
    for (int i = 0; i < 16384; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            asm volatile("v_cmp_lt_f32 s[10:11], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[10:11]" : "+v"(a[k]) : "v"(m), "v"(c) : "s10", "s11");
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 123.456f) out[0] = s;
}
FULL_KERNEL(full_bfi, "v_bfi_b32 %0, %1, %0, %2")
// one compare into vcc, then RUN selects on it (eight selects per iteration of k)
template <int RUN>
__global__ __launch_bounds__(256) void full_select_run(float *out, float m, float c) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = (float)(threadIdx.x + k);
    for (int i = 0; i < 16384; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k % RUN == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(m) : "vcc");
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c) : "vcc");
        }
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 123.456f) out[0] = s;
}
template <int RUN>
__global__ __launch_bounds__(256) void full_select_run_sgpr(float *out, float m, float c) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = (float)(threadIdx.x + k);
    for (int i = 0; i < 16384; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k % RUN == 0) asm volatile("v_cmp_lt_f32 s[10:11], %0, %1" : : "v"(a[k]), "v"(m) : "s10", "s11");
            asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a[k]) : "v"(c) : "s10", "s11");
        }
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 123.456f) out[0] = s;
}

template <class K>
static int full_chip(const char *name, K kernel, float *out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 0.0f;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kernel, dim3(256 * 8), dim3(256), 0, 0, out, 0.999f, 0.001f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double insts = 256.0 * 8 * 4 * 16384.0 * 8.0;
    printf("whole chip, %-16s %.3f ms  %.3f G wave instructions/s per SIMD\n", name, ms, insts / ms / 1e6 / 1024.0);
    return 0;
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
    if (full_chip("v_fma_f32", full_fma, out) || full_chip("v_mul_f32", full_mul, out) || full_chip("v_min_f32", full_min, out) ||
        full_chip("v_max3_f32", full_max3, out) || full_chip("v_cvt_f32_ubyte", full_cvt, out) ||
        full_chip("v_cndmask_b32", full_cndmask, out) || full_chip("v_cmp_lt_f32", full_cmp, out) ||
        full_chip("v_add_u32", full_addu, out) || full_chip("v_mov_b32", full_mov, out) ||
        full_chip("v_min_u32", full_minu, out) || full_chip("v_and_b32", full_and, out) || full_chip("v_bfe_u32", full_bfe, out) ||
        full_chip("v_perm_b32", full_perm, out) || full_chip("v_med3_f32", full_med3, out) ||
        full_chip("v_bfi_b32", full_bfi, out) || full_chip("selects: vcc x2", full_select_run<2>, out) ||
        full_chip("selects: vcc x4", full_select_run<4>, out) || full_chip("selects: vcc x8", full_select_run<8>, out) ||
        full_chip("selects: sgpr x2", full_select_run_sgpr<2>, out) || full_chip("selects: sgpr x4", full_select_run_sgpr<4>, out) ||
        full_chip("selects: sgpr x8", full_select_run_sgpr<8>, out) || full_chip("cmp+cndmask vcc /2", full_select_vcc, out) ||
        full_chip("cmp+cndmask sgpr /2", full_select_sgpr, out) || full_chip("v_cndmask sgpr", full_cndmask_s, out) || full_chip("v_cndmask fresh", full_cndmask_fresh, out))
        return 1;
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
