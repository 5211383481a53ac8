// Prints which XCD (HW_REG_XCC_ID) each of the first 32 workgroups of a launch ran on, and the spread of a big launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
}
int main() {
    const int n = 4096;
    unsigned *d, h[n];
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) printf("%u ", h[i]);
    int hist[16] = {0};
    for (int i = 0; i < n; ++i) hist[h[i] & 15]++;
    printf("\nhist:");
    for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
    printf("\n");
    return 0;
}
