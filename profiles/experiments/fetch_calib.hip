// What do the memory-side counters of gfx950 (FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ) report for the access patterns
// of the traversal kernels?  Every kernel below moves a byte count that is known exactly, in one launch with a name
// of its own, so that one `rocprofv3 --pmc <counter>` pass over this program gives counted units per known byte:
//
//   stream_read     16 bytes per lane, consecutive lanes on consecutive addresses, 4 GiB once through
//   stream_write    the same as stores, 1 GiB
//   gather<64>      every lane loads one 64-byte record (four 16-byte loads) at a hashed position of a 512 MiB table
//                   (twice the Infinity Cache; hashed over 4 GiB the same loads run at 0.4 of the rate, bound by
//                   address translation, which a scene of a few hundred MB is not):
//                   the pattern of a BVH node or primitive record fetch that misses every cache
//   gather<128>     every lane loads one 128-byte record (eight loads), 128-byte aligned
//   gather_pair     every lane loads the two 64-byte halves of one hashed 128-byte line as two separate records
//                   (what a layout that pairs items fetched together would do)
//
// `fetch_calib 1` also times the gathers with 32 .. 512 blocks of 1024 threads (sixteen waves, one block per CU while
// there are fewer blocks than CUs): if the rate of scattered line fetches is a property of the fabric / memory the
// rate stops growing before all CUs are in use, if it is a property of each CU's L1 miss path it grows with the CUs.
// (A stream restricted to a subset of the CUs with hipExtStreamCreateWithCUMask would say the same more directly;
// creating one with a partial mask crashed the process on the machines this ran on.)
//
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

struct alignas(16) F4 {
    float x, y, z, w;
};

#define CHECK(e)                                                                        \
    do {                                                                                \
        hipError_t e_ = (e);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                     \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

__global__ __launch_bounds__(256) void stream_read(const F4 *__restrict__ buf, size_t n_f4, float *out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_f4; i += (size_t)gridDim.x * 256) {
        const F4 v = buf[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc; // never true (the table holds zeros): keeps the loads
}

__global__ __launch_bounds__(256) void stream_write(F4 *__restrict__ buf, size_t n_f4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_f4; i += (size_t)gridDim.x * 256)
        buf[i] = F4{0.0f, 0.0f, 0.0f, 0.0f};
}

// BYTES = 64 or 128: one record per lane and iteration at a hashed, BYTES-aligned position.
template <int BYTES>
__global__ __launch_bounds__(1024) void gather(const F4 *__restrict__ buf, uint32_t mask64, int iters, float *out) {
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        uint32_t rec = (h >> 9) & mask64; // index of a 64-byte record
        if (BYTES == 128) rec &= ~1u;
        const F4 *p = buf + (size_t)rec * 4;
#pragma unroll
        for (int k = 0; k < BYTES / 16; ++k) {
            const F4 v = p[k];
            acc += v.x + v.y + v.z + v.w; // all four words, so that the load stays a 16-byte one
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

// Both halves of one hashed 128-byte line, as two records (eight loads in two groups of four).
__global__ __launch_bounds__(256) void gather_pair(const F4 *__restrict__ buf, uint32_t mask64, int iters, float *out) {
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t rec = (h >> 9) & mask64;
        const F4 *p = buf + (size_t)rec * 4;
        const F4 *q = buf + (size_t)(rec ^ 1u) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const F4 v = p[k];
            acc += v.x + v.y + v.z + v.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const F4 v = q[k];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

static float timed(hipStream_t s, void (*launch)(hipStream_t)) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch(s); // warm-up
    CHECK(hipEventRecord(e0, s));
    launch(s);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms;
}

static F4 *g_buf;
static float *g_out;
static const uint32_t RECORDS = 1u << 26; // 64-byte records in the 4 GiB buffer the streaming kernels run over
static const uint32_t TABLE = 1u << 23;   // 64-byte records of the 512 MiB the gathers hash into
static const int BLOCKS = 256 * 6, ITERS = 256;

int main(int argc, char **argv) {
    const int extra = argc > 1 ? atoi(argv[1]) : 0;
    setvbuf(stdout, nullptr, _IONBF, 0);
    CHECK(hipMalloc(&g_buf, (size_t)RECORDS * 64));
    CHECK(hipMemset(g_buf, 0, (size_t)RECORDS * 64));
    CHECK(hipMalloc(&g_out, 4096));
    CHECK(hipDeviceSynchronize());
    const double lanes = (double)BLOCKS * 256.0, recs = lanes * ITERS;
    float ms;

    ms = timed(0, [](hipStream_t s) { hipLaunchKernelGGL(stream_read, dim3(256 * 16), dim3(256), 0, s, g_buf, (size_t)RECORDS * 4, g_out); });
    printf("stream_read    bytes %.0f  %.3f ms  %.0f GB/s\n", (double)RECORDS * 64, ms, (double)RECORDS * 64 / ms / 1e6);
    ms = timed(0, [](hipStream_t s) { hipLaunchKernelGGL(stream_write, dim3(256 * 16), dim3(256), 0, s, g_buf, (size_t)RECORDS); });
    printf("stream_write   bytes %.0f  %.3f ms  %.0f GB/s\n", (double)RECORDS * 16, ms, (double)RECORDS * 16 / ms / 1e6);
    ms = timed(0, [](hipStream_t s) { hipLaunchKernelGGL(gather<64>, dim3(BLOCKS), dim3(256), 0, s, g_buf, TABLE - 1, ITERS, g_out); });
    printf("gather<64>     records %.0f useful bytes %.0f  %.3f ms  %.2f G records/s  %.0f GB/s useful\n", recs, recs * 64, ms, recs / ms / 1e6, recs * 64 / ms / 1e6);
    ms = timed(0, [](hipStream_t s) { hipLaunchKernelGGL(gather<128>, dim3(BLOCKS), dim3(256), 0, s, g_buf, TABLE - 1, ITERS, g_out); });
    printf("gather<128>    records %.0f useful bytes %.0f  %.3f ms  %.2f G records/s  %.0f GB/s useful\n", recs, recs * 128, ms, recs / ms / 1e6, recs * 128 / ms / 1e6);
    ms = timed(0, [](hipStream_t s) { hipLaunchKernelGGL(gather_pair, dim3(BLOCKS), dim3(256), 0, s, g_buf, TABLE - 1, ITERS, g_out); });
    printf("gather_pair    lines %.0f useful bytes %.0f  %.3f ms  %.2f G lines/s  %.0f GB/s useful\n", recs, recs * 128, ms, recs / ms / 1e6, recs * 128 / ms / 1e6);

    if (extra == 1) {
        static int n_blocks;
        for (int nb : {32, 64, 128, 192, 256, 512}) {
            n_blocks = nb;
            const double r = (double)nb * 1024.0 * ITERS;
            ms = timed(0, [](hipStream_t s) { hipLaunchKernelGGL(gather<64>, dim3(n_blocks), dim3(1024), 0, s, g_buf, TABLE - 1, ITERS, g_out); });
            printf("gather<64>  %3d blocks of 1024 threads  %.3f ms  %.2f G records/s\n", nb, ms, r / ms / 1e6);
            ms = timed(0, [](hipStream_t s) { hipLaunchKernelGGL(gather<128>, dim3(n_blocks), dim3(1024), 0, s, g_buf, TABLE - 1, ITERS, g_out); });
            printf("gather<128> %3d blocks of 1024 threads  %.3f ms  %.2f G records/s\n", nb, ms, r / ms / 1e6);
        }
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
