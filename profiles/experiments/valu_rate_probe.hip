// Issue cost of the VALU instructions the traversal kernel is made of, on gfx950: every SIMD runs 8 waves that each
// execute a long unrolled run of one instruction kind (4 independent chains); prints shader-clock cycles per wave
// instruction per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cycles, int iters) {
    float a = 1.0f + threadIdx.x, b = 0.5f + threadIdx.x, c = 0.25f, d = 2.0f;
    unsigned ua = threadIdx.x * 2654435761u, ub = ua ^ 0x9E3779B9u;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v pa = {a, b}, pb = {c, d};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) { a = __builtin_fmaf(a, 1.0001f, 0.5f); b = __builtin_fmaf(b, 0.9999f, 0.25f); c = __builtin_fmaf(c, 1.0002f, 0.125f); d = __builtin_fmaf(d, 0.9998f, 0.0625f); }
            if (KIND == 1) { asm volatile("v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte1 %2, %3\n v_cvt_f32_ubyte2 %4, %5\n v_cvt_f32_ubyte3 %6, %7" : "=v"(a), "+v"(ua), "=v"(b), "+v"(ub), "=v"(c), "+v"(ua), "=v"(d), "+v"(ub)); }
            if (KIND == 2) { asm volatile("v_max3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_min3_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (KIND == 3) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc"); }
            if (KIND == 4) { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (KIND == 5) { asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %0\n v_mul_lo_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %0" : "+v"(ua), "+v"(ub)); }
            if (KIND == 7) { asm volatile("v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %1, %2, %3, %0\n v_perm_b32 %2, %3, %0, %1\n v_perm_b32 %3, %0, %1, %2" : "+v"(ua), "+v"(ub), "+v"(a), "+v"(b)); }
            if (KIND == 8) { asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %2, %3, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %3, %0, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (KIND == 9) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0" : "+v"(pa), "+v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0" : "+v"(pa), "+v"(pb)); }
            if (KIND == 10) { asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (KIND == 11) { asm volatile("v_min_u32 %0, %0, %1\n v_max_u32 %1, %1, %0\n v_and_b32 %0, %0, %1\n v_bfe_u32 %1, %1, 8, 8" : "+v"(ua), "+v"(ub)); }
            if (KIND == 12) { asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_add_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (KIND == 13) { asm volatile("v_cvt_f32_f16 %0, %1\n v_cvt_f32_f16 %1, %2\n v_cvt_f32_f16 %2, %3\n v_cvt_f32_f16 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (KIND == 14) { asm volatile("v_cvt_f32_u32 %0, %1\n v_cvt_f32_u32 %1, %2\n v_cvt_f32_u32 %2, %3\n v_cvt_f32_u32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (KIND == 15) { asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_f32_u32_sdwa %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_u32_sdwa %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_u32_sdwa %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(a), "+v"(ua), "=v"(b), "+v"(ub), "=v"(c), "+v"(ua), "=v"(d), "+v"(ub)); }
            if (KIND == 6) { asm volatile("v_min_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_max_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (float)(ua + ub) + pa.x + pa.y + pb.x + pb.y;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, float *d_out, unsigned long long *d_cyc, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd; // 4 waves per block, 4 SIMDs per CU, 256 CUs
    const int iters = 4000;
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[64];
    hipMemcpy(h, d_cyc, sizeof h, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < 64; ++i) cyc += (double)h[i] / 64.0;
    const double instrs_per_wave = (double)iters * 64.0;
    // a SIMD runs waves_per_simd waves; elapsed counter ticks per instruction issued on that SIMD
    printf("%-34s waves/SIMD %d: %7.3f ms  %6.2f counter ticks per wave-instr per SIMD, %6.2f ns per wave-instr per SIMD\n", name,
           waves_per_simd, ms, cyc / (instrs_per_wave * waves_per_simd), ms * 1e6 / (instrs_per_wave * waves_per_simd));
}

int main() {
    float *d_out;
    unsigned long long *d_cyc;
    (void)hipMalloc(&d_out, sizeof(float) * 256 * 16 * 256);
    (void)hipMalloc(&d_cyc, sizeof(unsigned long long) * 256 * 16);
    for (int w : {8}) {
        run<0>("v_fma_f32", d_out, d_cyc, w);
        run<1>("v_cvt_f32_ubyteN", d_out, d_cyc, w);
        run<2>("v_max3/min3_f32", d_out, d_cyc, w);
        run<3>("v_cmp + v_cndmask", d_out, d_cyc, w);
        run<4>("v_rcp_f32", d_out, d_cyc, w);
        run<5>("v_mul_lo/hi_u32", d_out, d_cyc, w);
        run<6>("v_min/max_f32", d_out, d_cyc, w);
        run<7>("v_perm_b32", d_out, d_cyc, w);
        run<8>("v_fma_mix_f32 (f16 src0)", d_out, d_cyc, w);
        run<9>("v_pk_fma_f32 (per instr)", d_out, d_cyc, w);
        run<10>("v_fma_f32 3 vgpr", d_out, d_cyc, w);
        run<11>("v_min_u32/max_u32/and/bfe", d_out, d_cyc, w);
        run<12>("v_mul_f32/v_add_f32", d_out, d_cyc, w);
        run<13>("v_cvt_f32_f16", d_out, d_cyc, w);
        run<14>("v_cvt_f32_u32", d_out, d_cyc, w);
        run<15>("v_cvt_f32_u32_sdwa BYTE_n", d_out, d_cyc, w);
    }
    return 0;
}
