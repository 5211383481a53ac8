// kernels.h -- host-callable launchers of the HIP kernels (kernels.hip).  Plain C++ prototypes; no HIP types leak
// beyond `void *stream` (a hipStream_t).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "rt_device.h"

namespace rtk {

constexpr int BLOCK = 256;
constexpr int STATE_PLANES = 14; // per-path SoA planes, see RenderArgs::state
constexpr int PAYLOAD_PLANES = 11; // per-queue-entry SoA planes, see RenderArgs::payload
constexpr int ACCUM_PLANES = 22;
constexpr int HIT_PLANES = 5;
constexpr int RESOLVE_PLANES = 15;
constexpr uint32_t MAX_STACK = 160; // deepest tree the kernels accept (entries of traversal stack)
constexpr int PIXEL_BLOCK = 64;     // pixels whose samples are adjacent in path-id space (one wave of primary rays)
constexpr int SEGMENTS = 8;         // queue segments, one per XCD (each XCD has its own L2)
constexpr int COUNTER_STRIDE = 64;  // uint32 slots between two counters: every counter has a 256-byte line to itself
constexpr int LDS_STACK = 24;       // stack entries kept in LDS per lane; deeper entries spill to global memory
constexpr int QUEUE_STACK = 14;     // the same in the traversal kernel that defers its record tests (LDS also holds its queue)
constexpr int QUEUE_CAP = 128;      // record tests a wave of that kernel can have waiting (a power of two)
constexpr int SPILL_FROM = QUEUE_STACK < LDS_STACK ? QUEUE_STACK : LDS_STACK; // first stack entry that may spill

struct RayAoS {
    float org[3];
    float tnear;
    float dir[3];
    float tfar;
};
struct HitAoS {
    float t;
    float pos[3];
    float n[3];
    float uv[2];
    uint32_t geom_id, prim_id, mat_id;
};

// Arguments of one wavefront pass: `samples` consecutive samples of each of `n_pixels` pixels.
//
// Path ids keep all samples of a 64-pixel block together: pid = ((block * samples) + s) * 64 + p for pixel
// block * 64 + p, so n_paths = n_blocks * 64 * samples (pixels past n_pixels in the last block are never started).
// Work is split into SEGMENTS segments, block b belonging to segment b % SEGMENTS; every segment has its own queue
// region, counters and cursor.  A wave starts on the segment of the XCD it runs on and appends survivors to the
// segment they came from, so the rays one XCD's L2 sees at any time come from a few neighbouring pixel blocks, bounce
// after bounce; drained waves move on to the other segments so no XCD idles.
struct RenderArgs {
    rtd::SceneView scene;
    rtd::CameraView camera;
    const uint32_t *pixels; // n_pixels x (x, y)
    float *state;           // STATE_PLANES planes of n_paths floats, indexed by path id:
                            //   0-2 radiance, 3-5 normal, 6-8 position, 9-11 albedo, 12 z, 13 ray_depth
    float *hits;            // HIT_PLANES planes of SEGMENTS * seg_cap: t, Ng.xyz, geometry word of every queue entry's hit
    float *payload[2];      // the rays of a bounce in queue order, ping-pong by bounce parity: PAYLOAD_PLANES planes of
                            // SEGMENTS * seg_cap, segment x at x * seg_cap: 0 path id (bits), 1-3 origin, 4-6 direction,
                            // 7-9 beta, 10 ray_depth so far.  Bounce 0 has an implicit queue (every path in id order).
    uint32_t *counts;       // counts[(d * SEGMENTS + x) * COUNTER_STRIDE] = live paths of segment x entering bounce d
                            // (d >= 1); zeroed before the pass
    uint32_t *work;         // work[(d * SEGMENTS + x) * COUNTER_STRIDE] = next unclaimed position of that queue; zeroed
    uint32_t *spill;        // traversal stack overflow: (stack_need - SPILL_FROM) entries per launched thread
    uint32_t *visit_counters; // optional: [0] node visits, [1] primitive tests (64-bit pairs lo/hi), may be null
    uint32_t *pixel_done;     // optional: pixel_done[i] != 0 = pixel i is precise enough and takes no more samples in
                              // this call (its paths are neither generated nor accumulated); pixel_done[n_pixels]
                              // counts them.  Written by accumulate_kernel only.  May be null.
    uint32_t n_pixels;
    uint32_t n_blocks;    // ceil(n_pixels / PIXEL_BLOCK)
    uint32_t samples;     // samples of each pixel in this pass
    uint32_t seg_cap;     // queue entries per segment
    uint32_t n_paths;     // n_blocks * PIXEL_BLOCK * samples
    uint32_t claim;       // queue entries a wave claims with one atomic (a multiple of 64)
    int32_t refill_below; // a traversal wave flushes finished lanes and refills when fewer lanes than this are busy
    int32_t leaf_bias;    // a primitive-test step is run when leaf lanes + bias > node lanes
    int32_t tail_shade;   // tail_kernel shades once this many lanes of a wave wait for it (0: when they are the majority)
    int32_t queue_low;    // > 0: traversal defers record tests into a per-wave queue and also tests records when fewer
                          // lanes than this have an inner node to visit; 0: the wave votes on its phase every step
    float allowed_error;  // > 0: a pixel stops accumulating within this pass once its colour error is below it
    uint32_t sample_base; // first sample index of the pass
    uint32_t strata;      // floor(sqrt(spp))
    uint32_t flags;       // pixel sampler and filter kinds (rt_device.h RT_SAMPLER_*, RT_FILTER_* << 4)
    uint32_t max_depth;
    uint32_t integrator;
    uint64_t seed;
};

// d_spill must hold max(0, stack_need - LDS_STACK) * grid_blocks * BLOCK entries (stack_need = the tree's worst case;
// SPILL_FROM instead of LDS_STACK for the render launches).
int launch_intersect(const rtd::SceneView &scene, const RayAoS *d_rays, HitAoS *d_hits, size_t n, uint32_t *d_spill,
                     uint32_t grid_blocks, void *stream);
// Device address and size (in 32-bit words) of the statically allocated stack-overflow buffer on the current device;
// nonzero on failure.  The address differs from device to device: look it up once per device, with that device
// current.
int static_spill_address(uint32_t **out, size_t *words);
int launch_raygen(const RenderArgs &args, void *stream); // camera rays of a pass, before launch_extend(depth 0)
int launch_extend(const RenderArgs &args, uint32_t depth, uint32_t grid_blocks, void *stream);
// one thread per queue entry; max_segment_entries = the longest segment's entry count at this bounce
int launch_shade(const RenderArgs &args, uint32_t depth, uint32_t max_segment_entries, void *stream);
// every path alive at bounce depth0 (>= 1) to its end, by persistent waves that take the paths from the queue of that
// bounce (args.claim entries per atomic); d_spill must cover grid_blocks * BLOCK threads
int launch_tail(const RenderArgs &args, uint32_t depth0, uint32_t grid_blocks, void *stream);
int tail_blocks_per_cu(); // resident blocks of tail_kernel per CU (occupancy query)
// resident blocks per CU of extend_kernel, or of extend_queue_kernel (occupancy query)
int extend_blocks_per_cu(bool queued);
int launch_accumulate(const RenderArgs &args, float *d_accum, float count_inc, void *stream);
// d_lin[i] = row-major position of pixel d_xy[i] in the window (ox, oy, w, h); *d_bad |= 1 if a pixel lies outside
int launch_frame_index(const uint32_t *d_xy, uint32_t n, uint32_t ox, uint32_t oy, uint32_t w, uint32_t h,
                       uint32_t *d_lin, uint32_t *d_bad, void *stream);
// accumulator planes of n pixels (plane stride n) into the planes of a frame of n_dst pixels at positions d_lin
int launch_scatter_planes(const float *d_src, uint32_t n, const uint32_t *d_lin, float *d_dst, uint32_t n_dst,
                          void *stream);
int launch_resolve(const float *d_accum, uint32_t n_pixels, float *d_out, int srgb, void *stream);

} // namespace rtk
