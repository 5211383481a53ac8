// How fast does a CU's vector L1 (TCP) take the traversal kernel's access pattern on gfx950?  Every lane gathers
// 64-byte records at hashed positions of a buffer that fits L1 / L2 / only HBM, either as the kernels do (four
// 16-byte loads per lane, each instruction touching up to 64 different lines) or cooperatively (four neighbouring
// lanes load the four quarters of one record, so an instruction touches 16 lines).  Prints ns per record per CU
// and the implied line accesses per clock.  The "dependent" variants make each iteration's addresses depend on the
// data of the one before, as a traversal step does, so that a wave has one iteration's loads in flight and no more.   hipcc --offload-arch=gfx950 -O3 -o l1_rate_probe l1_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

struct alignas(16) F4 { float x, y, z, w; };

template <int MODE, bool DEP>
__global__ __launch_bounds__(256) void probe(const F4 *__restrict__ buf, unsigned mask, int iters, float *out) {
    __shared__ F4 xpose[256 * 5]; // 64 B per lane, padded to 80 B so that a lane's four 16-byte reads spread over the banks
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    const unsigned lane = threadIdx.x & 63u;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        if (MODE == 0) { // one record per lane, four 16-byte loads (every word is used: reading one word of each
                         // quarter lets the compiler shrink the loads to four 4-byte ones, which is MODE 6)
            const F4 *p = buf + (size_t)((h >> 8) & mask) * 4;
            F4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
            if (DEP) h += __float_as_uint(a.x) + __float_as_uint(d.w); // the buffer holds zeros: same sequence, real dependency
        } else if (MODE == 6) { // one record per lane, one word of each quarter: four 4-byte loads
            const F4 *p = buf + (size_t)((h >> 8) & mask) * 4;
            F4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 2) { // as MODE 1, then through LDS so that every lane ends up with its own whole record
            const unsigned wave_base = (threadIdx.x & ~63u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned hq = __shfl(h, (lane & ~3u) + k); // record of the quad's lane k
                const F4 *p = buf + (size_t)((hq >> 8) & mask) * 4 + (lane & 3u);
                xpose[(wave_base + (lane & ~3u) + k) * 5 + (lane & 3u)] = p[0];
            }
            const F4 *mine = &xpose[(wave_base + lane) * 5];
            F4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
            acc += a.x + b.y + c.z + d.w;
            if (DEP) h += __float_as_uint(a.x) + __float_as_uint(d.w);
        } else if (MODE == 5) { // eight lanes load one 128-byte line (two records): 8 lines per instruction
            unsigned hq = __shfl(h, lane & ~7u);
            const F4 *p = buf + (size_t)((hq >> 8) & mask & ~1u) * 4 + (lane & 7u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                F4 a = p[0];
                acc += a.x + a.y + a.z + a.w;
                hq = hq * 1664525u + 1013904223u;
                p = buf + (size_t)((hq >> 8) & mask & ~1u) * 4 + (lane & 7u);
            }
            h = hq; // the next iteration continues the sequence instead of repeating three of these four records
        } else { // one record per four lanes, one load each: the wave fetches 16 records per instruction
            unsigned hq = __shfl(h, lane & ~3u); // the quad's common record
            const F4 *p = buf + (size_t)((hq >> 8) & mask) * 4 + (lane & 3u);
#pragma unroll
            for (int k = 0; k < 4; ++k) { // four instructions = 64 records per wave, as MODE 0 fetches
                F4 a = p[0];
                acc += a.x + a.y + a.z + a.w; // all four words, so that the load stays a 16-byte one
                hq = hq * 1664525u + 1013904223u;
                p = buf + (size_t)((hq >> 8) & mask) * 4 + (lane & 3u);
            }
            h = hq;
            if (DEP) h += __float_as_uint(acc) & 0u ? 1u : 0u; // (this mode has no per-lane record; kept independent)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const size_t max_records = (size_t)1 << 26; // 4 GB
    F4 *buf;
    float *out;
    (void)hipMalloc(&buf, max_records * 64);
    (void)hipMemset(buf, 0, max_records * 64);
    const int blocks = 256 * 6;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    struct { const char *name; unsigned records; } sizes[] = {{"8 KB (L1)", 128}, {"1 MB (L2)", 16384}, {"64 MB (MALL)", 1u << 20}, {"256 MB", 1u << 22}, {"512 MB (HBM)", 1u << 23}, {"4 GB (HBM)", 1u << 26}};
    for (auto &s : sizes)
        for (int mode = 0; mode < 7; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL((probe<0, false>), dim3(blocks), dim3(256), 0, 0, buf, s.records - 1, iters, out);
                else if (mode == 1) hipLaunchKernelGGL((probe<1, false>), dim3(blocks), dim3(256), 0, 0, buf, s.records - 1, iters, out);
                else if (mode == 2) hipLaunchKernelGGL((probe<2, false>), dim3(blocks), dim3(256), 0, 0, buf, s.records - 1, iters, out);
                else if (mode == 3) hipLaunchKernelGGL((probe<0, true>), dim3(blocks), dim3(256), 0, 0, buf, s.records - 1, iters, out);
                else if (mode == 4) hipLaunchKernelGGL((probe<2, true>), dim3(blocks), dim3(256), 0, 0, buf, s.records - 1, iters, out);
                else if (mode == 5) hipLaunchKernelGGL((probe<5, false>), dim3(blocks), dim3(256), 0, 0, buf, s.records - 1, iters, out);
                else hipLaunchKernelGGL((probe<6, false>), dim3(blocks), dim3(256), 0, 0, buf, s.records - 1, iters, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // records fetched per CU: 6 blocks x 256 lanes x iters (x 1 in both modes: mode 1 fetches 4 per quad per iter)
            const double rec_per_cu = 6.0 * 256.0 * iters;
            const double ns_per_rec = ms * 1e6 / rec_per_cu;
            printf("%-14s %-22s %8.3f ms  %6.3f ns per 64-B record per CU  (%.2f records per clock at 2.4 GHz, %.0f GB/s per GPU)\n", s.name,
                   mode == 0 ? "4 loads per lane" : mode == 1 ? "1 load per lane, quads" : mode == 2 ? "quads + LDS transpose" : mode == 3 ? "4 loads/lane, dependent" : mode == 4 ? "quads + LDS, dependent" : mode == 5 ? "8 lanes per 128-B line" : "4 x 4-byte loads per lane", ms, ns_per_rec, 1.0 / (ns_per_rec * 2.4),
                   rec_per_cu * 256.0 * 64.0 / (ms * 1e-3) / 1e9);
        }
    return 0;
}
