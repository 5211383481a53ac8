// scene_build.h -- host-side scene assembly shared by the C ABI: primitive records, padded culling boxes, BVH build
// and the 64-byte primitive records (two triangles that share an edge, a lone triangle or a sphere) the kernels read.  No HIP dependency.
#pragma once

#include <functional>
#include <vector>

#include "bvh.h"
#include "rt_device.h"

namespace rts {

struct HostPrim {
    float a[3], b[3], c[3]; // triangle: v0, v1, v2; sphere: centre, (radius, 0, 0), unused
    uint32_t geom, prim, is_sphere;
};

// Transformer<Point>::apply (crates/rt/src/math/transform.rs:45-52) with glam's Quat * Vec3, in f32.
void apply_transform(const float translation[3], const float scale[3], const float rot[4], float p[3]);

// Pairs triangles into records, builds the BVH over the records (global primitive index = position in `prims`;
// the primitives of one geometry must be consecutive) and writes every record into the item array of the tree
// (bvh.items) at the place the tree's layout gives its leaf.  lo/hi receive the unpadded scene bounds; *n_records the
// number of records.  Returns false when the progress callback cancels.
bool build_scene(const std::vector<HostPrim> &prims, rtb::WideBvh &bvh, size_t *n_records, float lo[3], float hi[3],
                 const std::function<bool(double)> &progress);

} // namespace rts
