// bvh.h -- host-side acceleration structure build for the GPU kernels: binned-SAH binary BVH over padded primitive
// boxes, collapsed to 4-wide nodes in the quantised node layout of rt_device.h (48 bytes in a 64-byte item), and laid out together with the
// 64-byte slots of the primitive records in ONE array of 64-byte items (a 128-byte line, the unit the memory system
// fetches, holds two of them).
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

// Placement in the item array: the children of a node (inner nodes and the records of its leaves) take consecutive
// items in slot order, groups breadth-first, every group starting on a 128-byte line; a node names its children by the
// index of the first.  (A placement that put every node next to its largest child, and records two by two, fetched
// 6-7 % fewer lines from beyond L2 per incoherent ray on a 5 M-triangle mesh and gained no time; it is gone with the
// per-child references it needed.)
struct WideBvh {
    std::vector<rtd::F4> items;        // n_items x RT_NODE_F4: nodes in place (root = item 0), record slots zeroed
    std::vector<uint32_t> prim_order;  // leaf slot -> index of the input primitive (a primitive that spatial splits cut has several leaves)
    std::vector<uint32_t> record_item; // leaf slot -> item the primitive's record goes to
    uint32_t n_items = 0;              // nodes + record slots + padding
    uint32_t n_nodes = 0;
    uint32_t stack_need = 0; // worst-case traversal stack entries
    uint32_t depth = 0;
    float sah_cost = 0.0f;
    uint32_t spatial_splits = 0; // inner nodes of the binary tree that cut references at a plane
};

// Padded bounds of the part of primitive `prim` that lies inside `region`; false when no part does.  Must be
// conservative (contain every point of the primitive in the region, padded like the primitive's own box) and
// monotonic (a smaller region gives a box inside the larger region's).
typedef std::function<bool(uint32_t prim, const Box &region, Box &out)> ClipFn;

// boxes: one padded box per primitive.  progress(amount in [0,1]) returning false cancels; build() then returns
// false.  threads = worker threads for subtree builds (0 = hardware concurrency, capped at 16).  With a clip function
// the builder may cut primitives at planes (spatial splits, what Embree's RTC_BUILD_QUALITY_HIGH does for the
// reference, crates/rt/src/aggregate/embree.rs:33-34); a cut primitive has one leaf per part, all naming the same
// input primitive.  RTMI_SPATIAL=0 builds without.
bool build(const std::vector<Box> &boxes, WideBvh &out, const std::function<bool(double)> &progress,
           unsigned threads = 0, const ClipFn &clip = ClipFn());

} // namespace rtb
