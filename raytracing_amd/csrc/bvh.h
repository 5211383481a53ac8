// bvh.h -- host-side acceleration structure build for the GPU kernels: binned-SAH binary BVH over padded primitive
// boxes, collapsed to 4-wide nodes in the 64-byte quantised node layout of rt_device.h, and laid out together with the
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

// How nodes and record slots are placed in the item array.
//   BFS    all nodes breadth-first, then all record slots in leaf order: the inner children of a node are neighbours,
//          so siblings often share a line.  The default.
//   PAIRS  every 128-byte line holds a node and its largest child (an inner node or a record: the child a ray that
//          reaches the node is most likely to visit), or two leftover records of one parent; lines breadth-first.
//          Measured on a 5 M-triangle mesh: 6-7 % fewer lines fetched from beyond L2 per incoherent ray, 1 % more L2
//          requests, 14 % more memory, and no time gained (the traversal kernel is bound by instruction issue and
//          by the L1 address path at that point), hence not the default.
enum class Layout { BFS, PAIRS };

struct WideBvh {
    std::vector<rtd::F4> items;        // n_items x RT_NODE_F4: nodes in place (root = item 0), record slots zeroed
    std::vector<uint32_t> prim_order;  // leaf slot -> index of the input primitive
    std::vector<uint32_t> record_item; // leaf slot -> item the primitive's record goes to
    uint32_t n_items = 0;              // nodes + record slots + padding
    uint32_t n_nodes = 0;
    uint32_t stack_need = 0; // worst-case traversal stack entries
    uint32_t depth = 0;
    float sah_cost = 0.0f;
};

// boxes: one padded box per primitive.  progress(amount in [0,1]) returning false cancels; build() then returns
// false.  threads = worker threads for subtree builds (0 = hardware concurrency, capped at 16).
bool build(const std::vector<Box> &boxes, WideBvh &out, const std::function<bool(double)> &progress,
           unsigned threads = 0, Layout layout = Layout::BFS);

} // namespace rtb
