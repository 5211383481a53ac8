// Does VALU issue time on gfx950 depend on which lanes of a wave64 are active?  Runs a long chain of FMAs under
// different exec masks on a full grid and prints the time of each: if sparse masks are no faster, a step of the
// traversal kernel costs the same however few of its lanes work, and only instruction count and lane utilisation matter.
//   hipcc --offload-arch=gfx950 -O3 -o exec_mask_probe exec_mask_probe.hip && ./exec_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void probe(float *out, unsigned long long mask, int iters) {
    const unsigned lane = threadIdx.x & 63u;
    float a = 1.0f + lane, b = 0.5f, c = 0.25f, d = 2.0f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                a = __builtin_fmaf(a, 1.0001f, 0.5f);
                b = __builtin_fmaf(b, 0.9999f, 0.25f);
                c = __builtin_fmaf(c, 1.0002f, 0.125f);
                d = __builtin_fmaf(d, 0.9998f, 0.0625f);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

int main() {
    float *d_out;
    const int blocks = 256 * 8;
    hipMalloc(&d_out, sizeof(float) * blocks * 256);
    struct { const char *name; unsigned long long mask; } cases[] = {
        {"all 64 lanes", ~0ull},
        {"lanes 0-31", 0xFFFFFFFFull},
        {"lanes 0-15", 0xFFFFull},
        {"lanes 0-15 and 32-47", 0x0000FFFF0000FFFFull},
        {"every 2nd lane (32 active)", 0x5555555555555555ull},
        {"every 4th lane (16 active)", 0x1111111111111111ull},
        {"one lane", 1ull},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &c : cases) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d_out, c.mask, 2000);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d_out, c.mask, 20000);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.3f ms\n", c.name, ms);
    }
    return 0;
}
