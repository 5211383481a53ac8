// kernels.h -- host-callable launchers of the HIP kernels (kernels.hip).  Plain C++ prototypes; no HIP types leak
// beyond `void *stream` (a hipStream_t).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "rt_device.h"

namespace rtk {

constexpr int BLOCK = 256;
constexpr int STATE_PLANES = 23; // per-path SoA planes, see RenderArgs::state
constexpr int ACCUM_PLANES = 22;
constexpr int RESOLVE_PLANES = 15;
constexpr uint32_t MAX_STACK = 160; // deepest tree the kernels accept (entries of traversal stack)
constexpr int LDS_STACK = 24;       // stack entries kept in LDS per lane; deeper entries spill to global memory

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

// Arguments of one wavefront pass: `samples` consecutive samples of each of `n_pixels` pixels, path id
// pid = s * n_pixels + pixel, n_paths = samples * n_pixels.
struct RenderArgs {
    rtd::SceneView scene;
    rtd::CameraView camera;
    const uint32_t *pixels; // n_pixels x (x, y)
    float *state;           // STATE_PLANES planes of n_paths floats:
                            //   0-2 origin, 3-5 direction, 6-8 beta, 9-11 radiance, 12-14 normal, 15-17 position,
                            //   18-20 albedo, 21 z, 22 ray_depth
    uint32_t *queue[2];     // live path ids, ping-pong by bounce parity
    uint32_t *counts;       // counts[d] = live paths entering bounce d (d >= 1); zeroed before the pass
    uint32_t *work;         // work[d] = next unclaimed queue position of bounce d; zeroed before the pass
    uint32_t *spill;        // traversal stack overflow: (stack_need - LDS_STACK) entries per launched thread
    uint32_t *visit_counters; // optional: [0] node visits, [1] primitive tests (64-bit pairs lo/hi), may be null
    uint32_t n_pixels;
    uint32_t n_paths;
    uint32_t sample_base; // first sample index of the pass
    uint32_t strata;      // floor(sqrt(spp))
    uint32_t max_depth;
    uint32_t integrator;
    uint64_t seed;
};

// d_spill must hold max(0, stack_need - LDS_STACK) * grid_blocks * BLOCK entries (stack_need = the tree's worst case).
int launch_intersect(const rtd::SceneView &scene, const RayAoS *d_rays, HitAoS *d_hits, size_t n, uint32_t *d_spill,
                     uint32_t grid_blocks, void *stream);
int launch_bounce(const RenderArgs &args, uint32_t depth, uint32_t grid_blocks, void *stream);
int launch_accumulate(const RenderArgs &args, float *d_accum, uint32_t samples, float count_inc, void *stream);
int launch_resolve(const float *d_accum, uint32_t n_pixels, float *d_out, int srgb, void *stream);

} // namespace rtk
