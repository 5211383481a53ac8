// kernels.hip -- gfx950 kernels of the wavefront path tracer.
//
//   intersect_kernel   batch closest hit (Shape::intersection_full for n rays)
//   bounce_kernel      one bounce of every live path: [camera ray at bounce 0] -> closest hit -> shade -> next ray,
//                      then wave-level compaction of the survivors into the next bounce's queue
//   accumulate_kernel  adds a pass's samples to the per-pixel accumulators, in sample order
//   resolve_kernel     accumulators -> the seven output channels
//
// One ray per lane.  The traversal stack lives in LDS, one column per lane (stack[entry][lane], bank = lane), so a
// push or pop is one conflict-free ds_write_b32 / ds_read_b32.  Waves claim 64 queue entries at a time with one
// atomic on a per-bounce cursor and publish survivors with one atomic per wave (ballot + popcount give each lane its
// rank).  There is no barrier anywhere: waves never wait for each other.
#include <hip/hip_runtime.h>

#include "kernels.h"

using namespace rtd;

namespace rtk {

// Per-lane traversal stack: the first LDS_STACK entries live in LDS, deeper ones (rare: the tree's worst case is far
// above the typical depth) spill to a per-thread column of a global buffer sized from the committed tree.
struct LaneStack {
    uint32_t *col;  // &lds[thread of block]; entry e is at col[e * BLOCK]
    uint32_t *ovf;  // &spill[global thread]; entry LDS_STACK + e is at ovf[e * ovf_stride]
    size_t ovf_stride;
    int sp;
    __device__ __forceinline__ void push(uint32_t v) {
        if (sp < LDS_STACK)
            col[sp * BLOCK] = v;
        else
            ovf[(size_t)(sp - LDS_STACK) * ovf_stride] = v;
        ++sp;
    }
    __device__ __forceinline__ uint32_t pop() {
        --sp;
        return sp < LDS_STACK ? col[sp * BLOCK] : ovf[(size_t)(sp - LDS_STACK) * ovf_stride];
    }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
};

__device__ __forceinline__ LaneStack make_stack(uint32_t *lds, uint32_t *spill) {
    LaneStack st;
    st.col = lds + threadIdx.x;
    st.ovf_stride = (size_t)gridDim.x * BLOCK;
    st.ovf = spill + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    st.sp = 0;
    return st;
}

__global__ __launch_bounds__(BLOCK) void intersect_kernel(SceneView scene, const RayAoS *__restrict__ rays,
                                                          HitAoS *__restrict__ hits, uint64_t n, uint32_t *spill) {
    __shared__ uint32_t lds[LDS_STACK * BLOCK];
    LaneStack st = make_stack(lds, spill);
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        const F4 *rp = reinterpret_cast<const F4 *>(rays + i);
        F4 r0 = rp[0], r1 = rp[1];
        V3 o = mk(r0.x, r0.y, r0.z), d = mk(r1.x, r1.y, r1.z);
        st.sp = 0;
        Closest best;
        closest_hit(scene, o, d, r0.w, r1.w, st, best, (uint32_t *)nullptr, (uint32_t *)nullptr);
        HitInfo h;
        finish_hit(scene, o, d, best, h);
        F4 *hp = reinterpret_cast<F4 *>(hits + i);
        F4 h0, h1, h2;
        h0.x = h.t;
        h0.y = h.pos.x;
        h0.z = h.pos.y;
        h0.w = h.pos.z;
        h1.x = h.n.x;
        h1.y = h.n.y;
        h1.z = h.n.z;
        h1.w = h.u;
        h2.x = h.v;
        h2.y = __builtin_bit_cast(float, h.geom_id);
        h2.z = __builtin_bit_cast(float, h.prim_id);
        h2.w = __builtin_bit_cast(float, h.mat_id);
        hp[0] = h0;
        hp[1] = h1;
        hp[2] = h2;
    }
}

// Lanes refill themselves: a lane whose ray has finished traversal shades it, publishes the continuation and takes the
// next queue entry, while the other lanes of the wave keep their rays.  To keep both halves efficient the wave only
// leaves the traversal loop when fewer than RenderArgs::refill_below lanes are still traversing (or the queue is
// drained and nothing is left), so traversal runs with most lanes busy and shading runs for a batch of lanes at a time.

// XCC_ID hardware register (bits 3:0): the XCD this wave runs on.  Used for locality only, never for correctness.
__device__ __forceinline__ uint32_t xcd_of_wave() {
    return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & (uint32_t)(SEGMENTS - 1);
}

__global__ __launch_bounds__(BLOCK) void bounce_kernel(RenderArgs a, uint32_t depth) {
    __shared__ uint32_t lds[LDS_STACK * BLOCK];
    LaneStack st = make_stack(lds, a.spill);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    const size_t P = a.n_paths;
    float *__restrict__ S = a.state;
    uint32_t nodes = 0, tests = 0;
    const bool count_visits = a.visit_counters != nullptr;
    const uint32_t per_block = a.samples * (uint32_t)PIXEL_BLOCK; // paths of one pixel block
    const uint32_t home = xcd_of_wave();

    for (uint32_t turn = 0; turn < (uint32_t)SEGMENTS; ++turn) {
        const uint32_t seg = (home + turn) & (uint32_t)(SEGMENTS - 1);
        // blocks seg, seg + SEGMENTS, ... belong to this segment
        const uint32_t seg_blocks = a.n_blocks > seg ? (a.n_blocks - seg + (uint32_t)SEGMENTS - 1u) / (uint32_t)SEGMENTS : 0u;
        const uint32_t n_in =
            (depth == 0u) ? seg_blocks * per_block : a.counts[(size_t)(depth * SEGMENTS + seg) * COUNTER_STRIDE];
        const uint32_t *__restrict__ q_in = a.queue[depth & 1u] + (size_t)seg * a.seg_cap;
        uint32_t *__restrict__ q_out = a.queue[(depth + 1u) & 1u] + (size_t)seg * a.seg_cap;
        uint32_t *cursor = &a.work[(size_t)(depth * SEGMENTS + seg) * COUNTER_STRIDE];
        uint32_t *out_count = &a.counts[(size_t)((depth + 1u) * SEGMENTS + seg) * COUNTER_STRIDE];
        uint32_t c_next = 0, c_end = 0; // wave-uniform: the part of the queue this wave has claimed and not yet used

        bool has_ray = false;
        bool drained = n_in == 0u; // wave-uniform: the segment has no unclaimed entries left
        uint32_t pid = 0, x = 0, y = 0, sample = 0;
        V3 org = mk(0.0f, 0.0f, 0.0f), dir = mk(0.0f, 0.0f, 1.0f), beta = mk(0.0f, 0.0f, 0.0f);
        Trav tr;
        tr.cur = RT_EMPTY;

        for (;;) {
            // ---- shade the lanes whose traversal is complete
            bool alive = false;
            if (has_ray && tr.cur == RT_EMPTY) {
                HitInfo h;
                finish_hit(a.scene, org, dir, tr.best, h);
                VertexOut vo;
                bool cont = shade_vertex(a.scene, h, a.integrator, a.seed, x, y, sample, depth, org, dir, beta, vo);
                if (depth == 0u) {
                    S[12 * P + pid] = vo.normal.x;
                    S[13 * P + pid] = vo.normal.y;
                    S[14 * P + pid] = vo.normal.z;
                    S[15 * P + pid] = vo.position.x;
                    S[16 * P + pid] = vo.position.y;
                    S[17 * P + pid] = vo.position.z;
                    S[18 * P + pid] = vo.albedo.x;
                    S[19 * P + pid] = vo.albedo.y;
                    S[20 * P + pid] = vo.albedo.z;
                    S[21 * P + pid] = vo.z;
                    S[22 * P + pid] = 0.0f + vo.t_add;
                } else if (h.hit && vo.t_add != 0.0f) {
                    S[22 * P + pid] = S[22 * P + pid] + vo.t_add;
                }
                if (cont && depth + 1u == a.max_depth) {
                    // the next vertex would be at the depth limit: it contributes beta * 0 (pathtracing.rs:24-26)
                    vo.radiance = mul(beta, mk(0.0f, 0.0f, 0.0f));
                    cont = false;
                }
                if (cont) {
                    S[0 * P + pid] = org.x;
                    S[1 * P + pid] = org.y;
                    S[2 * P + pid] = org.z;
                    S[3 * P + pid] = dir.x;
                    S[4 * P + pid] = dir.y;
                    S[5 * P + pid] = dir.z;
                    S[6 * P + pid] = beta.x;
                    S[7 * P + pid] = beta.y;
                    S[8 * P + pid] = beta.z;
                    alive = true;
                } else {
                    S[9 * P + pid] = vo.radiance.x;
                    S[10 * P + pid] = vo.radiance.y;
                    S[11 * P + pid] = vo.radiance.z;
                }
                has_ray = false;
            }
            // ---- publish survivors: one atomic per wave, lanes ranked by the ballot
            const unsigned long long amask = __ballot(alive);
            if (amask != 0ull) {
                uint32_t obase = 0;
                if (lane == 0) obase = atomicAdd(out_count, (uint32_t)__popcll(amask));
                obase = __builtin_amdgcn_readfirstlane(obase);
                if (alive) q_out[obase + (uint32_t)__popcll(amask & below)] = pid;
            }
            // ---- refill idle lanes from the wave's claimed range; claim the next a.claim entries with one atomic
            // when the range is used up
            if (c_next == c_end && !drained) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(cursor, a.claim);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n_in) {
                    drained = true;
                } else {
                    c_next = base;
                    c_end = min(base + a.claim, n_in);
                    if (c_end == n_in) drained = true;
                }
            }
            if (c_next != c_end) {
                const unsigned long long nmask = __ballot(!has_ray);
                const uint32_t i = c_next + (uint32_t)__popcll(nmask & below);
                const bool take = !has_ray && i < c_end;
                c_next = min(c_end, c_next + (uint32_t)__popcll(nmask));
                if (take) {
                    bool valid = true;
                    uint32_t pix;
                    if (depth == 0u) {
                        // entry i of the segment: local block i / per_block, then sample-major inside the block
                        const uint32_t lb = i / per_block, rem = i - lb * per_block;
                        const uint32_t blk = lb * (uint32_t)SEGMENTS + seg;
                        const uint32_t s_local = rem / (uint32_t)PIXEL_BLOCK, p = rem % (uint32_t)PIXEL_BLOCK;
                        pix = blk * (uint32_t)PIXEL_BLOCK + p;
                        pid = blk * per_block + rem;
                        sample = a.sample_base + s_local;
                        valid = pix < a.n_pixels;
                    } else {
                        pid = q_in[i];
                        const uint32_t blk = pid / per_block, rem = pid - blk * per_block;
                        pix = blk * (uint32_t)PIXEL_BLOCK + rem % (uint32_t)PIXEL_BLOCK;
                        sample = a.sample_base + rem / (uint32_t)PIXEL_BLOCK;
                    }
                    if (valid) {
                        x = a.pixels[2 * pix];
                        y = a.pixels[2 * pix + 1];
                        if (depth == 0u) {
                            camera_ray(a.camera, a.strata, a.seed, x, y, sample, org, dir);
                            beta = mk(1.0f, 1.0f, 1.0f);
                        } else {
                            org = mk(S[0 * P + pid], S[1 * P + pid], S[2 * P + pid]);
                            dir = mk(S[3 * P + pid], S[4 * P + pid], S[5 * P + pid]);
                            beta = mk(S[6 * P + pid], S[7 * P + pid], S[8 * P + pid]);
                        }
                        st.sp = 0;
                        trav_begin(a.scene, org, dir, INFINITY, tr);
                        has_ray = true;
                    }
                }
            }
            const bool more_to_claim = !drained || c_next != c_end;
            if (__ballot(has_ray) == 0ull) {
                if (!more_to_claim) break;
                continue; // only padding entries were taken: take more
            }
            // ---- traverse while enough lanes are busy (all the way once nothing is left to take).  Every step the wave
            // runs the one phase -- inner-node test or primitive test -- that most of its busy lanes are waiting for,
            // so each step executes with at least half of the busy lanes.
            const int keep_going = more_to_claim ? a.refill_below : 1;
            for (;;) {
                const bool busy = has_ray && tr.cur != RT_EMPTY;
                const bool at_leaf = busy && (tr.cur & RT_LEAF) != 0u;
                const bool at_node = busy && !at_leaf;
                const int n_node = (int)__popcll(__ballot(at_node));
                const int n_leaf = (int)__popcll(__ballot(at_leaf));
                if (n_node + n_leaf < keep_going) break;
                if (n_leaf == 0 || (n_node > 0 && n_node >= n_leaf + a.leaf_bias)) {
                    if (at_node) {
                        if (count_visits) ++nodes;
                        trav_node_step(a.scene, RT_TNEAR, tr, st);
                    }
                } else {
                    if (at_leaf) {
                        if (count_visits) ++tests;
                        trav_leaf_step(a.scene, org, dir, RT_TNEAR, tr, st);
                    }
                }
            }
        }
    }
    if (count_visits) {
        // wave-level sums, then one 64-bit atomic pair per wave
        for (int off = 32; off > 0; off >>= 1) {
            nodes += __shfl_down(nodes, off);
            tests += __shfl_down(tests, off);
        }
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(a.visit_counters), (unsigned long long)nodes);
            atomicAdd(reinterpret_cast<unsigned long long *>(a.visit_counters) + 1, (unsigned long long)tests);
        }
    }
}

// crates/rt/src/math/stat.rs:45-52
__device__ __forceinline__ float variance_of(float count, float sum, float sqsum) {
    if (count <= 2.0f) return INFINITY;
    return (sqsum - sum * sum / count) / (count - 1.0f);
}

// VarianceSeries::is_precise_enough (crates/rt/src/math/stat.rs:57-78) with the Student table of stat.rs:92-95
__device__ __forceinline__ bool channel_precise(float count, float sum, float sqsum, float abs_err) {
    const float student[16] = {2.131f, 2.120f, 2.110f, 2.101f, 2.093f, 2.086f, 2.080f, 2.074f,
                               2.069f, 2.064f, 2.060f, 2.056f, 2.052f, 2.048f, 2.045f, 2.042f};
    const float df = count - 1.0f;
    if (df < 15.0f) return false;
    const int idx = (int)(df < 30.0f ? df : 30.0f) - 15;
    float t = student[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) t = (idx == k) ? student[k] : t;
    const float err = t * sqrtf(variance_of(count, sum, sqsum));
    return err < abs_err;
}

// RaySeries::add_sample (crates/rt/src/renderer.rs:63-82, crates/rt/src/math/stat.rs:24-28,165-168) for the `samples`
// samples of each pixel, in sample order, filter weight 1 (crates/rt/src/filter.rs:30).
__global__ __launch_bounds__(BLOCK) void accumulate_kernel(RenderArgs a, float *__restrict__ accum, float count_inc) {
    const size_t P = a.n_paths;
    const size_t N = a.n_pixels;
    const float *__restrict__ S = a.state;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (size_t)gridDim.x * BLOCK) {
        float acc[ACCUM_PLANES];
#pragma unroll
        for (int k = 0; k < ACCUM_PLANES; ++k) acc[k] = accum[(size_t)k * N + i];
        const size_t block_base = (i / PIXEL_BLOCK) * (size_t)a.samples * PIXEL_BLOCK + i % PIXEL_BLOCK;
        for (uint32_t s = 0; s < a.samples; ++s) {
            const size_t pid = block_base + (size_t)s * PIXEL_BLOCK;
            const float w = 1.0f;
            const float col[3] = {S[9 * P + pid], S[10 * P + pid], S[11 * P + pid]};
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                acc[ch] += col[ch];
                acc[3 + ch] += col[ch] * col[ch];
                acc[6 + ch] = acc[6 + ch] + w * col[ch];
            }
            acc[9] += w;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[10 + k] = acc[10 + k] + S[(size_t)(12 + k) * P + pid];
            acc[19] += S[21 * P + pid];
            acc[20] += S[22 * P + pid];
            acc[21] += count_inc;
            // crates/rt-cli/src/executor.rs:244-249: the pixel is done for this batch once all channels are precise
            if (a.allowed_error > 0.0f && channel_precise(acc[21], acc[0], acc[3], a.allowed_error) &&
                channel_precise(acc[21], acc[1], acc[4], a.allowed_error) &&
                channel_precise(acc[21], acc[2], acc[5], a.allowed_error))
                break;
        }
#pragma unroll
        for (int k = 0; k < ACCUM_PLANES; ++k) accum[(size_t)k * N + i] = acc[k];
    }
}

// crates/rt/src/color/colorspace.rs:71-80
__device__ __forceinline__ float srgb_from_linear(float linear) {
    if (isnan(linear)) return 0.0f;
    linear = linear < 0.0f ? 0.0f : (linear > 1.0f ? 1.0f : linear);
    if (linear < 0.0031308f) return 12.92f * linear;
    return 1.055f * powf(linear, 1.0f / 2.4f) - 0.055f;
}

// RaySeries::as_pixelresult (crates/rt/src/renderer.rs:37-61) [+ OutputBuffers::convert, output/mod.rs:46-58]
__global__ __launch_bounds__(BLOCK) void resolve_kernel(const float *__restrict__ accum, uint32_t n_pixels,
                                                        float *__restrict__ out, int srgb) {
    const size_t N = n_pixels;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (size_t)gridDim.x * BLOCK) {
        float A[ACCUM_PLANES];
#pragma unroll
        for (int k = 0; k < ACCUM_PLANES; ++k) A[k] = accum[(size_t)k * N + i];
        float O[RESOLVE_PLANES];
        const float count = A[21];
        const float inv = 1.0f / count;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            O[0 + c] = inv * A[10 + c];
            O[3 + c] = inv * A[13 + c];
            O[6 + c] = inv * A[16 + c];
            O[9 + c] = (A[9] == 0.0f) ? A[6 + c] : A[6 + c] / A[9];
        }
        const float vr = variance_of(count, A[0], A[3]);
        const float vg = variance_of(count, A[1], A[4]);
        const float vb = variance_of(count, A[2], A[5]);
        O[12] = sqrtf(vr * vr + vg * vg + vb * vb);
        O[13] = inv * A[19];
        O[14] = inv * A[20];
        if (srgb) {
#pragma unroll
            for (int c = 0; c < 12; ++c) O[c] = srgb_from_linear(O[c]);
        }
#pragma unroll
        for (int k = 0; k < RESOLVE_PLANES; ++k) out[(size_t)k * N + i] = O[k];
    }
}

// ---------------------------------------------------------------------------------------------- launchers

static int check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static uint32_t blocks_for(size_t n, uint32_t cap) {
    size_t b = (n + BLOCK - 1) / BLOCK;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (uint32_t)b;
}

int launch_intersect(const SceneView &scene, const RayAoS *d_rays, HitAoS *d_hits, size_t n, uint32_t *d_spill,
                     uint32_t grid_blocks, void *stream) {
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t grid = blocks_for(n, grid_blocks);
    hipLaunchKernelGGL(intersect_kernel, dim3(grid), dim3(BLOCK), 0, s, scene, d_rays, d_hits, (uint64_t)n, d_spill);
    return check_launch();
}

int launch_bounce(const RenderArgs &args, uint32_t depth, uint32_t grid_blocks, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bounce_kernel, dim3(grid_blocks), dim3(BLOCK), 0, s, args, depth);
    return check_launch();
}

int launch_accumulate(const RenderArgs &args, float *d_accum, float count_inc, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const uint32_t grid = blocks_for(args.n_pixels, 256u * 16u);
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid), dim3(BLOCK), 0, s, args, d_accum, count_inc);
    return check_launch();
}

int launch_resolve(const float *d_accum, uint32_t n_pixels, float *d_out, int srgb, void *stream) {
    if (n_pixels == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t grid = blocks_for(n_pixels, 256u * 16u);
    hipLaunchKernelGGL(resolve_kernel, dim3(grid), dim3(BLOCK), 0, s, d_accum, n_pixels, d_out, srgb);
    return check_launch();
}

} // namespace rtk
