// bvh.h -- host-side acceleration structure build for the GPU kernels: binned-SAH binary BVH over padded primitive
// boxes, collapsed to 4-wide nodes and flattened breadth-first into the 64-byte quantised node layout of rt_device.h.
// Stands in for what rtcCommitScene does inside Embree for the reference (crates/rt/src/aggregate/embree.rs:63-69).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "rt_device.h"

namespace rtb {

struct Box {
    float lo[3], hi[3];
};

struct WideBvh {
    std::vector<rtd::F4> nodes;       // RT_NODE_F4 entries per node, root first, breadth-first
    std::vector<uint32_t> prim_order; // slot -> index of the input primitive
    uint32_t n_nodes = 0;
    uint32_t stack_need = 0; // worst-case traversal stack entries
    uint32_t depth = 0;
    float sah_cost = 0.0f;
};

// boxes: one padded box per primitive.  progress(amount in [0,1]) returning false cancels; build() then returns
// false.  threads = worker threads for subtree builds (0 = hardware concurrency, capped at 16).
bool build(const std::vector<Box> &boxes, WideBvh &out, const std::function<bool(double)> &progress,
           unsigned threads = 0);

} // namespace rtb
