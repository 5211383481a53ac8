// kernels.hip -- gfx950 kernels of the wavefront path tracer.
//
//   intersect_kernel   batch closest hit (Shape::intersection_full for n rays)
//   raygen_kernel      camera rays of every path of a pass
//   extend_kernel      closest hit of every live path's ray at one bounce (persistent waves, lanes refill themselves)
//   extend_queue_kernel  the same with the record tests deferred into a per-wave queue (used for the first bounces)
//   shade_kernel       scatter / terminate every path from its hit record, compact the survivors into the next
//                      bounce's queue
//   tail_kernel        every path still alive once a bounce no longer fills the GPU, each run to its end in one launch
//   accumulate_kernel  adds a pass's samples to the per-pixel accumulators, in sample order
//   resolve_kernel     accumulators -> the seven output channels
//
// One ray per lane.  The traversal stack lives in LDS, one column per lane (stack[entry][lane], bank = lane), so a
// push or pop is one conflict-free ds_write_b32 / ds_read_b32.  Traversal waves claim batches of queue entries with
// one atomic on a per-bounce cursor and never wait for each other.
#include <hip/hip_runtime.h>

#include "kernels.h"

using namespace rtd;

namespace rtk {

// Per-lane traversal stack: the first LDS_STACK entries live in LDS, deeper ones (rare: the tree's worst case is far
// above the typical depth) spill to a per-thread column of a global buffer sized from the committed tree.
typedef __attribute__((address_space(3))) uint32_t lds_u32; // typed LDS: accesses stay ds_read / ds_write

template <int N> // N = entries kept in LDS
struct LaneStackT {
    lds_u32 *col;    // &lds[thread of block]; entry e is at col[e * BLOCK]
    uint32_t *spill; // wave-uniform base; entry N + e of a thread is at spill[e * threads + global thread]
    int sp;
    bool shallow; // wave-uniform: no lane of the wave is within 4 entries of the LDS part's end (see note_depth)
    // uniform base + 32-bit byte offset (the overflow buffer stays below 4 GiB: MAX_STACK entries of a resident grid)
    __device__ __forceinline__ uint32_t *deep(int e) const {
        const uint32_t threads = gridDim.x * (uint32_t)BLOCK;
        const uint32_t off = ((uint32_t)(e - N) * threads + (blockIdx.x * (uint32_t)BLOCK + threadIdx.x)) << 2;
        return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(spill) + off);
    }
    __device__ __forceinline__ void push(uint32_t v) {
        if (shallow || sp < N)
            col[sp * BLOCK] = v;
        else
            *deep(sp) = v;
        ++sp;
    }
    __device__ __forceinline__ uint32_t pop() {
        --sp;
        return (shallow || sp < N) ? col[sp * BLOCK] : *deep(sp);
    }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
    // Call with all lanes of the wave active, before a step that pushes at most three entries: the common case of a
    // shallow stack then skips the per-lane spill checks.
    __device__ __forceinline__ void note_depth() { shallow = __ballot(sp > N - 4) == 0ull; }
};
typedef LaneStackT<LDS_STACK> LaneStack;
typedef LaneStackT<QUEUE_STACK> QueueStack; // of the kernels that keep a queue of record tests in LDS as well

template <int N>
__device__ __forceinline__ LaneStackT<N> make_stack_of(uint32_t *lds, uint32_t *spill) {
    LaneStackT<N> st;
    st.col = (lds_u32 *)lds + threadIdx.x;
    st.spill = spill;
    st.sp = 0;
    st.shallow = false;
    return st;
}
__device__ __forceinline__ LaneStack make_stack(uint32_t *lds, uint32_t *spill) { return make_stack_of<LDS_STACK>(lds, spill); }

// Number of set bits of a wave mask below this lane.
__device__ __forceinline__ uint32_t rank_in(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Batch closest hit (Shape::intersection_full for n rays).  Same lane scheduling as extend_kernel below: a lane whose
// ray has finished waits until a quarter of the wave is idle, then the idle lanes write their hits and take the next
// rays of the wave's share, and every step the wave runs the phase (inner node or primitive record) most of its busy
// lanes wait for.  A wave's share is every `waves`-th chunk of 64 consecutive rays, not one contiguous slice: batches
// come in the caller's order (camera rays row by row, say), the cost of a ray varies over the batch by orders of
// magnitude, and a contiguous slice per wave leaves the waves of the cheap regions idle while the others work.
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(6, 6))) void intersect_kernel(
    SceneView scene, const RayAoS *__restrict__ rays, HitAoS *__restrict__ hits, uint64_t n, uint32_t *spill) {
    __shared__ uint32_t lds[LDS_STACK * BLOCK];
    LaneStack st = make_stack(lds, spill);
    const uint64_t waves = (uint64_t)gridDim.x * (BLOCK / 64);
    const uint64_t wave = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t chunks = (n + 63) / 64;
    uint64_t next = 0; // wave-uniform position in the share: chunk next / 64 of this wave, ray next % 64 of it
    const uint64_t end = chunks > wave ? (chunks - wave + waves - 1) / waves * 64 : 0;
    bool has_ray = false;
    uint64_t index = 0;
    float tnear = 0.0f;
    V3 org = mk(0.0f, 0.0f, 0.0f), dir = mk(0.0f, 0.0f, 1.0f);
    Trav tr;
    tr.cur = RT_EMPTY;
    for (;;) {
        const bool busy = has_ray && tr.cur != RT_EMPTY;
        const int n_busy = (int)__popcll(__ballot(busy));
        if (n_busy < 48) {
            if (has_ray && !busy) {
                HitInfo h;
                finish_hit(scene, org, dir, tr.best, h);
                F4 *hp = reinterpret_cast<F4 *>(hits + index);
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
                has_ray = false;
            }
            if (next < end) {
                const unsigned long long nmask = __ballot(!has_ray);
                const uint64_t v = next + rank_in(nmask);
                const uint64_t i = ((v >> 6) * waves + wave) * 64 + (v & 63);
                next = next + (uint64_t)__popcll(nmask) < end ? next + (uint64_t)__popcll(nmask) : end;
                if (!has_ray && v < end && i < n) {
                    const F4 *rp = reinterpret_cast<const F4 *>(rays + i);
                    const F4 r0 = rp[0], r1 = rp[1];
                    org = mk(r0.x, r0.y, r0.z);
                    dir = mk(r1.x, r1.y, r1.z);
                    tnear = r0.w;
                    index = i;
                    st.sp = 0;
                    trav_begin(scene, org, dir, r1.w, tr);
                    has_ray = true;
                }
            }
            if (__ballot(has_ray) == 0ull) {
                if (next >= end) break;
                continue;
            }
        }
        st.note_depth();
        const bool at_leaf = has_ray && tr.cur != RT_EMPTY && (tr.cur & RT_LEAF) != 0u;
        const bool at_node = has_ray && tr.cur != RT_EMPTY && !at_leaf;
        const int n_node = (int)__popcll(__ballot(at_node));
        const int n_leaf = (int)__popcll(__ballot(at_leaf));
        if (n_leaf == 0 || (n_node > 0 && n_node >= n_leaf)) {
            if (at_node) trav_node_step(scene, tnear, tr, st);
        } else {
            if (at_leaf) trav_leaf_step(scene, org, dir, tnear, tr, st);
        }
    }
}

// Stack-overflow buffer of the batch closest-hit entry point that must not allocate: a file-scope array every GPU
// holds a copy of from the moment the library is loaded there until the process ends.  Host code must never use the
// array's name as an address (on the host it is only a stand-in for the symbol): static_spill_address() asks the
// runtime for the device address, once per device, and every kernel argument goes through that pointer.
constexpr size_t STATIC_SPILL_WORDS = (size_t)8 << 20; // 32 MiB
static __device__ __align__(256) uint32_t scratch[STATIC_SPILL_WORDS];

int static_spill_address(uint32_t **out, size_t *words) {
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) return e != hipSuccess ? (int)e : -1;
    *out = static_cast<uint32_t *>(p);
    *words = STATIC_SPILL_WORDS;
    return 0;
}

// Element idx of a plane: uniform plane base + 32-bit byte offset (a plane is at most 4 GiB).  Planes are streamed (an
// element is written by one kernel and read once by the next), so they are accessed with the non-temporal hint and
// leave the caches to the tree (measured: 0.7 ms per frame off the tail kernel, nothing elsewhere).
__device__ __forceinline__ float ldp(const float *plane, uint32_t idx) {
    return __builtin_nontemporal_load(reinterpret_cast<const float *>(reinterpret_cast<const char *>(plane) + (idx << 2)));
}
__device__ __forceinline__ void stp(float *plane, uint32_t idx, float v) {
    __builtin_nontemporal_store(v, reinterpret_cast<float *>(reinterpret_cast<char *>(plane) + (idx << 2)));
}

// XCC_ID hardware register (bits 3:0): the XCD this wave runs on.  Used for locality only, never for correctness.
__device__ __forceinline__ uint32_t xcd_of_wave() {
    return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & (uint32_t)(SEGMENTS - 1);
}

// Entry i of segment seg at bounce `depth` -> path id, pixel and sample.  At bounce 0 the queue is implicit (every
// path of the segment's pixel blocks, sample-major inside a block; entries of pixels past n_pixels are padding).
struct Entry {
    uint32_t pid, pix, sample;
    bool valid;
};
__device__ __forceinline__ Entry queue_entry(const RenderArgs &a, uint32_t depth, uint32_t seg, uint32_t i,
                                             const float *__restrict__ q_in) {
    const uint32_t per_block = a.samples * (uint32_t)PIXEL_BLOCK;
    Entry e;
    if (depth == 0u) {
        const uint32_t lb = i / per_block, rem = i - lb * per_block;
        const uint32_t blk = lb * (uint32_t)SEGMENTS + seg;
        e.pix = blk * (uint32_t)PIXEL_BLOCK + rem % (uint32_t)PIXEL_BLOCK;
        e.pid = blk * per_block + rem;
        e.sample = a.sample_base + rem / (uint32_t)PIXEL_BLOCK;
        e.valid = e.pix < a.n_pixels && !(a.pixel_done != nullptr && a.pixel_done[e.pix] != 0u);
    } else {
        e.pid = __builtin_bit_cast(uint32_t, q_in[i]);
        const uint32_t blk = e.pid / per_block, rem = e.pid - blk * per_block;
        e.pix = blk * (uint32_t)PIXEL_BLOCK + rem % (uint32_t)PIXEL_BLOCK;
        e.sample = a.sample_base + rem / (uint32_t)PIXEL_BLOCK;
        e.valid = true;
    }
    return e;
}

__device__ __forceinline__ uint32_t segment_entries(const RenderArgs &a, uint32_t depth, uint32_t seg) {
    if (depth == 0u) {
        const uint32_t seg_blocks = a.n_blocks > seg ? (a.n_blocks - seg + (uint32_t)SEGMENTS - 1u) / (uint32_t)SEGMENTS : 0u;
        return seg_blocks * a.samples * (uint32_t)PIXEL_BLOCK;
    }
    return a.counts[(size_t)(depth * SEGMENTS + seg) * COUNTER_STRIDE];
}

// Wave-uniform cursor of a traversal wave over the queue of one bounce: the segment being drained (the wave starts on
// the segment of the XCD it runs on and moves on to the others once that one is used up) and the part of that
// segment's queue the wave has claimed and not yet handed to its lanes.  Every member must be called with all lanes of
// the wave active.
struct QueueCursor {
    uint32_t home, turn, seg, n_in, c_next, c_end;
    bool drained; // the segment has no unclaimed entries left
    __device__ __forceinline__ void begin(const RenderArgs &a, uint32_t depth) {
        home = xcd_of_wave();
        turn = 0;
        seg = home;
        n_in = segment_entries(a, depth, seg);
        c_next = c_end = 0;
        drained = n_in == 0u;
    }
    // When the claimed range is used up: the next segment that has entries left, and a.claim of them with one atomic.
    __device__ __forceinline__ void claim(const RenderArgs &a, uint32_t depth) {
        while (c_next == c_end && drained && turn + 1u < (uint32_t)SEGMENTS) {
            ++turn;
            seg = (home + turn) & (uint32_t)(SEGMENTS - 1);
            n_in = segment_entries(a, depth, seg);
            drained = n_in == 0u;
        }
        if (c_next == c_end && !drained) {
            uint32_t base = 0;
            if (rank_in(~0ull) == 0u) base = atomicAdd(&a.work[(size_t)(depth * SEGMENTS + seg) * COUNTER_STRIDE], a.claim);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= n_in) {
                drained = true;
            } else {
                c_next = base;
                c_end = min(base + a.claim, n_in);
                if (c_end == n_in) drained = true;
            }
        }
    }
    // Deals the next claimed entries to the lanes that are idle, in lane order; true for a lane that got entry i.
    __device__ __forceinline__ bool take(bool idle, uint32_t &i) {
        if (c_next == c_end) return false;
        const unsigned long long nmask = __ballot(idle);
        i = c_next + rank_in(nmask);
        const bool got = idle && i < c_end;
        c_next = min(c_end, c_next + (uint32_t)__popcll(nmask));
        return got;
    }
    __device__ __forceinline__ bool finished() const { return c_next == c_end && drained && turn + 1u >= (uint32_t)SEGMENTS; }
};

// Hit record of the queue entry at plane offset `slot`: geometry word (RT_EMPTY = miss) and, for a hit, distance and
// geometric normal (shade_kernel reads those two for hits only, so a miss leaves them as they are).
__device__ __forceinline__ void store_hit(float *__restrict__ hits, size_t H, uint32_t slot, float t, V3 ng, uint32_t code) {
    stp(hits + 4 * H, slot, __builtin_bit_cast(float, code));
    if (code != RT_EMPTY) {
        stp(hits + 0 * H, slot, t);
        stp(hits + 1 * H, slot, ng.x);
        stp(hits + 2 * H, slot, ng.y);
        stp(hits + 3 * H, slot, ng.z);
    }
}

// Origin of every camera ray: the centre of the lens (camera_ray in rt_device.h; camera.rs:62-85 moves only the point
// the ray aims at over the aperture).
__device__ __forceinline__ V3 camera_origin(const RenderArgs &a) { return mk(a.camera.lens[0], a.camera.lens[1], a.camera.lens[2]); }

// Camera rays of every path of a pass, in the implicit queue order of bounce 0, into the bounce's payload planes.
// Padding entries (pixels past n_pixels, pixels that take no more samples) get a zero direction, which no camera ray
// has.
__global__ __launch_bounds__(BLOCK) void raygen_kernel(RenderArgs a) {
    const uint32_t seg = blockIdx.x % (uint32_t)SEGMENTS;
    const uint32_t i = (blockIdx.x / (uint32_t)SEGMENTS) * (uint32_t)BLOCK + threadIdx.x;
    if (i >= segment_entries(a, 0u, seg)) return;
    const size_t H = (size_t)SEGMENTS * a.seg_cap;
    float *__restrict__ Q = a.payload[0];
    const uint32_t pos = seg * a.seg_cap + i;
    const Entry e = queue_entry(a, 0u, seg, i, nullptr);
    V3 org = mk(0.0f, 0.0f, 0.0f), dir = mk(0.0f, 0.0f, 0.0f);
    if (e.valid) camera_ray(a.camera, a.strata, a.flags, a.seed, a.pixels[2 * e.pix], a.pixels[2 * e.pix + 1], e.sample, org, dir);
    // every camera ray starts at the centre of the lens (camera_ray): the kernels of bounce 0 take the origin from the
    // camera, and only the direction travels through memory
    stp(Q + 4 * H, pos, dir.x);
    stp(Q + 5 * H, pos, dir.y);
    stp(Q + 6 * H, pos, dir.z);
}

// Closest hit of every ray of one bounce.  Waves are persistent: a lane whose ray has finished writes its hit record
// and takes the next queue entry while the other lanes of the wave keep their rays, so the traversal loop runs with
// most lanes busy.  Finished lanes are flushed and refilled in batches (when fewer than RenderArgs::refill_below lanes
// are busy), since that code runs for the idle lanes only.  Every step the wave runs the one phase -- inner-node test
// or primitive test -- that most of its busy lanes are waiting for.
//
// Hit record of queue entry i of segment x, at plane offset x * seg_cap + i: distance (+inf = miss), geometric normal
// (not normalised), geometry word.  Rays come from the bounce's payload planes, in queue order.
// The queue of bounce 0 has padding entries (zero direction) that are skipped.
__global__ __launch_bounds__(BLOCK) void extend_kernel(RenderArgs a, uint32_t depth) {
    __shared__ uint32_t lds[LDS_STACK * BLOCK];
    LaneStack st = make_stack(lds, a.spill);
    const bool primary = depth == 0u;
    const bool count = a.visit_counters != nullptr; // tally node visits and primitive tests
    const size_t H = (size_t)SEGMENTS * a.seg_cap;
    float *__restrict__ hits = a.hits;
    const float *__restrict__ Q = a.payload[depth & 1u]; // this bounce's rays, in queue order
    uint32_t nodes = 0, tests = 0;
    QueueCursor cur;
    cur.begin(a, depth);

    bool has_ray = false;
    uint32_t slot_out = 0; // where this lane's hit record goes
    V3 org = mk(0.0f, 0.0f, 0.0f), dir = mk(0.0f, 0.0f, 1.0f);
    Trav tr;
    tr.cur = RT_EMPTY;

    for (;;) {
        const bool busy = has_ray && tr.cur != RT_EMPTY;
        const int n_busy = (int)__popcll(__ballot(busy));
        if (n_busy < a.refill_below) {
            // ---- flush the finished lanes
            if (has_ray && !busy) {
                float t = INFINITY;
                V3 ng = mk(0.0f, 0.0f, 0.0f);
                uint32_t code = RT_EMPTY, geom_first;
                if (tr.best.gidx != RT_EMPTY) {
                    t = tr.best.t;
                    ng = candidate_normal(a.scene, tr.best.slot, org, dir, t, code, geom_first);
                }
                store_hit(hits, H, slot_out, t, ng, code);
                has_ray = false;
            }
            // ---- refill the idle lanes from the wave's claimed range
            cur.claim(a, depth);
            uint32_t i;
            if (cur.take(!has_ray, i)) {
                const uint32_t pos = cur.seg * a.seg_cap + i;
                org = primary ? camera_origin(a) : mk(ldp(Q + 1 * H, pos), ldp(Q + 2 * H, pos), ldp(Q + 3 * H, pos));
                dir = mk(ldp(Q + 4 * H, pos), ldp(Q + 5 * H, pos), ldp(Q + 6 * H, pos));
                if (!primary || dir.x != 0.0f || dir.y != 0.0f || dir.z != 0.0f) {
                    slot_out = pos;
                    st.sp = 0;
                    trav_begin(a.scene, org, dir, INFINITY, tr);
                    has_ray = true;
                }
            }
            if (__ballot(has_ray) == 0ull) {
                // nothing in flight: done when every segment is used up, otherwise take more
                if (cur.finished()) break;
                continue;
            }
        }
        // ---- one traversal step
        st.note_depth();
        const bool at_leaf = has_ray && tr.cur != RT_EMPTY && (tr.cur & RT_LEAF) != 0u;
        const bool at_node = has_ray && tr.cur != RT_EMPTY && !at_leaf;
        const int n_node = (int)__popcll(__ballot(at_node));
        const int n_leaf = (int)__popcll(__ballot(at_leaf));
        if (n_leaf == 0 || (n_node > 0 && n_node >= n_leaf + a.leaf_bias)) {
            if (at_node) {
                if (count) ++nodes;
                trav_node_step(a.scene, RT_TNEAR, tr, st);
            }
        } else {
            if (at_leaf) {
                if (count) ++tests;
                trav_leaf_step(a.scene, org, dir, RT_TNEAR, tr, st);
            }
        }
    }
    if (count) {
        // wave-level sums, then one 64-bit atomic pair per wave
        for (int off = 32; off > 0; off >>= 1) {
            nodes += __shfl_down(nodes, off);
            tests += __shfl_down(tests, off);
        }
        if (rank_in(~0ull) == 0u) {
            atomicAdd(reinterpret_cast<unsigned long long *>(a.visit_counters), (unsigned long long)nodes);
            atomicAdd(reinterpret_cast<unsigned long long *>(a.visit_counters) + 1, (unsigned long long)tests);
        }
    }
}

// extend_kernel with the record tests deferred.  A lane that reaches a leaf does not wait for the wave to run a record
// phase: it appends (record, lane) to a queue the wave keeps in LDS and goes on with its next stack entry.  A record
// phase takes the oldest 64 queue entries, one per lane whichever lane they came from: the tester reads the owner's
// ray from LDS (the owner itself traverses with the reciprocals it keeps in registers), tests the record and merges its
// candidate into the owner's best hit in LDS with a 64-bit minimum over (distance bits, primitive index) -- the order
// closest hits are decided in anyway, so deferring changes no result.  Record phases therefore run with every lane busy and inner-node
// steps never wait for lanes parked at a leaf; the price is that inner-node steps cull against a best distance that
// lags behind by the queued tests (a few percent more visits).  A lane is flushed once its traversal is over and all
// its queue entries have been tested.
struct ExtendQueueLds {
    uint32_t stack[QUEUE_STACK * BLOCK];
    unsigned long long best[BLOCK]; // (bits of t) << 32 | global primitive index; t > 0, so the order is (t, index)
    uint32_t best_slot[BLOCK];      // Closest::slot of that candidate
    float2 ray[BLOCK][3];           // origin and direction of the lane's ray
    uint32_t slot_out[BLOCK];       // where the lane's hit record goes
    uint32_t queue[BLOCK / 64][QUEUE_CAP]; // per wave: owner lane << 26 | record slot (a scene has fewer than 2^26 records)
};
static_assert(RT_MAX_ITEMS <= (1u << 26), "a queue entry keeps the record slot in 26 bits");

// The queue of one wave (wave-uniform counters; entry k is at entries[k % QUEUE_CAP]) and the three things a wave does
// with it.  Every member must be called with all lanes of the wave active.
struct WaveQueue {
    uint32_t *entries;
    uint32_t head, tail; // entries tested / appended so far
    __device__ __forceinline__ uint32_t length() const { return tail - head; }
    // Lanes at a leaf append it (while there is room) and go on with their next stack entry.  last_post = `tail` right
    // after the lane's latest entry: the lane has tests outstanding while `head` is below it.
    __device__ __forceinline__ void post(bool has_ray, uint32_t lane, Trav &tr, QueueStack &st, uint32_t &last_post) {
        const bool at_leaf = has_ray && tr.cur != RT_EMPTY && (tr.cur & RT_LEAF) != 0u;
        const unsigned long long lmask = __ballot(at_leaf);
        if (lmask == 0ull) return;
        const uint32_t room = (uint32_t)QUEUE_CAP - (tail - head);
        const uint32_t r = rank_in(lmask);
        if (at_leaf && r < room) {
            const uint32_t first = tr.cur & 0x0FFFFFFFu, more = (tr.cur >> 28) & 7u;
            entries[(tail + r) & (uint32_t)(QUEUE_CAP - 1)] = (lane << 26) | first;
            last_post = tail + r + 1u;
            if (more != 0u)
                tr.cur = RT_LEAF | ((more - 1u) << 28) | (first + 1u);
            else
                tr.cur = st.empty() ? RT_EMPTY : st.pop();
        }
        tail += min((uint32_t)__popcll(lmask), room);
    }
    __device__ __forceinline__ bool outstanding(uint32_t last_post) const { return (int32_t)(last_post - head) > 0; }
    // The oldest entries, one per lane: the owner's ray from LDS, the record test, the candidate merged into the
    // owner's best hit; the lane whose candidate is the owner's best after every lane's minimum records its slot.
    __device__ __forceinline__ void test(ExtendQueueLds &lds, const SceneView &scene, uint32_t wave_base, uint32_t lane) {
        const uint32_t take = min(length(), 64u);
        if (lane < take) {
            const uint32_t e = entries[(head + lane) & (uint32_t)(QUEUE_CAP - 1)];
            const uint32_t owner = wave_base + (e >> 26);
            const float2 r0 = lds.ray[owner][0], r1 = lds.ray[owner][1], r2 = lds.ray[owner][2];
            const V3 o = mk(r0.x, r0.y, r1.x), d = mk(r1.y, r2.x, r2.y);
            Closest c;
            c.t = INFINITY;
            c.gidx = RT_EMPTY;
            c.slot = 0;
            c.u = c.v = 0.0f;
            test_prim(scene, e & 0x03FFFFFFu, o, d, RT_TNEAR, c);
            if (c.gidx != RT_EMPTY) {
                const unsigned long long key = ((unsigned long long)__builtin_bit_cast(uint32_t, c.t) << 32) | c.gidx;
                atomicMin(&lds.best[owner], key);
                if (lds.best[owner] == key) lds.best_slot[owner] = c.slot;
            }
        }
        head += take;
    }
    // A record phase is due: a wave's worth of tests is waiting, the queue is about to overflow, too few lanes have an
    // inner node to visit, or nothing else is left to do.
    __device__ __forceinline__ bool due(uint32_t n_node, uint32_t n_leaf, uint32_t low) const {
        const uint32_t q_len = length();
        return q_len >= 64u || q_len + n_leaf > (uint32_t)QUEUE_CAP || (n_node == 0u && n_leaf == 0u && q_len > 0u) ||
               (n_node < q_len && n_node < low);
    }
};

// Inner-node step of a lane whose ray and best hit live in LDS: the best distance as the queued tests have left it,
// and -origin / direction recomputed from the origin (the same operations trav_begin does: three registers fewer to
// carry through the record phases).
__device__ __forceinline__ void queued_node_step(ExtendQueueLds &lds, const SceneView &scene, uint32_t tid, Trav &tr, QueueStack &st) {
    tr.best.t = __builtin_bit_cast(float, reinterpret_cast<const uint32_t *>(&lds.best[tid])[1]);
    const float2 r0 = lds.ray[tid][0], r1 = lds.ray[tid][1];
    tr.nox = -(r0.x * tr.ix);
    tr.noy = -(r0.y * tr.iy);
    tr.noz = -(r1.x * tr.iz);
    trav_node_step(scene, RT_TNEAR, tr, st);
}

// A lane takes a ray: ray and an empty best hit into LDS, traversal state at the root.
__device__ __forceinline__ void queued_begin(ExtendQueueLds &lds, const SceneView &scene, uint32_t tid, V3 org, V3 dir, Trav &tr, QueueStack &st) {
    st.sp = 0;
    trav_begin(scene, org, dir, INFINITY, tr);
    lds.ray[tid][0] = make_float2(org.x, org.y);
    lds.ray[tid][1] = make_float2(org.z, dir.x);
    lds.ray[tid][2] = make_float2(dir.y, dir.z);
    lds.best[tid] = ((unsigned long long)__builtin_bit_cast(uint32_t, INFINITY) << 32) | RT_EMPTY;
}

// The closest hit the queued tests have left for a lane: distance (+inf = none), geometric normal, geometry word.
__device__ __forceinline__ void queued_result(ExtendQueueLds &lds, const SceneView &scene, uint32_t tid, float &t, V3 &ng, uint32_t &code, V3 &org, V3 &dir) {
    const unsigned long long key = lds.best[tid];
    const float2 r0 = lds.ray[tid][0], r1 = lds.ray[tid][1], r2 = lds.ray[tid][2];
    org = mk(r0.x, r0.y, r1.x);
    dir = mk(r1.y, r2.x, r2.y);
    t = INFINITY;
    ng = mk(0.0f, 0.0f, 0.0f);
    code = RT_EMPTY;
    if ((uint32_t)key != RT_EMPTY) {
        uint32_t geom_first;
        t = __builtin_bit_cast(float, (uint32_t)(key >> 32));
        ng = candidate_normal(scene, lds.best_slot[tid], org, dir, t, code, geom_first);
    }
}

__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(6, 6))) void extend_queue_kernel(RenderArgs a, uint32_t depth) {
    __shared__ ExtendQueueLds lds;
    QueueStack st = make_stack_of<QUEUE_STACK>(lds.stack, a.spill);
    const bool primary = depth == 0u;
    const size_t H = (size_t)SEGMENTS * a.seg_cap;
    float *__restrict__ hits = a.hits;
    const float *__restrict__ Q = a.payload[depth & 1u];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave_base = __builtin_amdgcn_readfirstlane(tid & ~63u); // wave-uniform, kept in a scalar register
    WaveQueue wq{lds.queue[wave_base >> 6], 0u, 0u};
    QueueCursor cur;
    cur.begin(a, depth);
    uint32_t last_post = 0;
    bool has_ray = false;
    Trav tr;
    tr.cur = RT_EMPTY;

    for (;;) {
        const bool live = has_ray && (tr.cur != RT_EMPTY || wq.outstanding(last_post));
        const int n_live = (int)__popcll(__ballot(live));
        if (n_live < a.refill_below) {
            if (has_ray && !live) {
                float t;
                V3 ng, org, dir;
                uint32_t code;
                queued_result(lds, a.scene, tid, t, ng, code, org, dir);
                store_hit(hits, H, lds.slot_out[tid], t, ng, code);
                has_ray = false;
            }
            cur.claim(a, depth);
            uint32_t i;
            if (cur.take(!has_ray, i)) {
                const uint32_t pos = cur.seg * a.seg_cap + i;
                const V3 org = primary ? camera_origin(a) : mk(ldp(Q + 1 * H, pos), ldp(Q + 2 * H, pos), ldp(Q + 3 * H, pos));
                const V3 dir = mk(ldp(Q + 4 * H, pos), ldp(Q + 5 * H, pos), ldp(Q + 6 * H, pos));
                if (!primary || dir.x != 0.0f || dir.y != 0.0f || dir.z != 0.0f) {
                    lds.slot_out[tid] = pos;
                    queued_begin(lds, a.scene, tid, org, dir, tr, st);
                    last_post = wq.head;
                    has_ray = true;
                }
            }
            if (__ballot(has_ray) == 0ull) {
                if (cur.finished()) break;
                continue;
            }
        }
        st.note_depth();
        wq.post(has_ray, lane, tr, st, last_post);
        const bool at_leaf = has_ray && tr.cur != RT_EMPTY && (tr.cur & RT_LEAF) != 0u;
        const bool at_node = has_ray && tr.cur != RT_EMPTY && !at_leaf;
        const uint32_t n_node = (uint32_t)__popcll(__ballot(at_node));
        const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
        if (wq.due(n_node, n_leaf, (uint32_t)a.queue_low)) {
            wq.test(lds, a.scene, wave_base, lane);
        } else if (at_node) {
            queued_node_step(lds, a.scene, tid, tr, st);
        }
    }
}

// Shades every hit record of one bounce, one queue entry per thread: scatter or terminate the path, store the first
// bounce's outputs, append the survivors to the next bounce's queue (one atomic per block).  Block b works on segment
// b % SEGMENTS, so the blocks an XCD receives read what that XCD's waves mostly wrote.
__global__ __launch_bounds__(BLOCK) void shade_kernel(RenderArgs a, uint32_t depth) {
    __shared__ uint32_t wave_count[BLOCK / 64];
    __shared__ uint32_t block_base;
    const uint32_t seg = blockIdx.x % (uint32_t)SEGMENTS;
    const uint32_t i = (blockIdx.x / (uint32_t)SEGMENTS) * (uint32_t)BLOCK + threadIdx.x;
    const uint32_t n_in = segment_entries(a, depth, seg);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t P = a.n_paths;
    const size_t H = (size_t)SEGMENTS * a.seg_cap;
    float *__restrict__ S = a.state;
    const float *__restrict__ hits = a.hits;
    const float *__restrict__ Q = a.payload[depth & 1u];
    float *__restrict__ Qn = a.payload[(depth + 1u) & 1u];
    const uint32_t pos = seg * a.seg_cap + i;

    bool alive = false;
    uint32_t pid = 0;
    V3 org, dir, beta;
    float ray_depth = 0.0f;
    if (i < n_in) {
        const Entry e = queue_entry(a, depth, seg, i, Q + (size_t)seg * a.seg_cap);
        pid = e.pid;
        if (e.valid) {
            org = depth == 0u ? camera_origin(a) : mk(ldp(Q + 1 * H, pos), ldp(Q + 2 * H, pos), ldp(Q + 3 * H, pos));
            dir = mk(ldp(Q + 4 * H, pos), ldp(Q + 5 * H, pos), ldp(Q + 6 * H, pos));
            beta = mk(1.0f, 1.0f, 1.0f);
            if (depth != 0u) {
                beta = mk(ldp(Q + 7 * H, pos), ldp(Q + 8 * H, pos), ldp(Q + 9 * H, pos));
                ray_depth = ldp(Q + 10 * H, pos);
            }
            const uint32_t code = __builtin_bit_cast(uint32_t, ldp(hits + 4 * H, pos));
            HitInfo h;
            h.u = h.v = 0.0f;
            h.prim_id = RT_EMPTY;
            if (code == RT_EMPTY) {
                h.hit = false;
                h.t = INFINITY;
                h.pos = mk(0.0f, 0.0f, 0.0f);
                h.n = mk(0.0f, 0.0f, 0.0f);
                h.geom_id = RT_EMPTY;
                h.mat_id = 0;
            } else {
                h.hit = true;
                h.t = ldp(hits + 0 * H, pos);
                h.pos = mk(org.x + h.t * dir.x, org.y + h.t * dir.y, org.z + h.t * dir.z);
                h.n = normalize_or_zero(mk(ldp(hits + 1 * H, pos), ldp(hits + 2 * H, pos), ldp(hits + 3 * H, pos)));
                h.geom_id = code & RT_GEOM_MASK;
                h.mat_id = h.geom_id < a.scene.n_geoms ? a.scene.geom_material[h.geom_id] : 0u;
                if (h.mat_id >= a.scene.n_materials) h.mat_id = 0u;
            }
            VertexOut vo;
            bool cont = shade_vertex(a.scene, h, a.integrator, a.seed, a.pixels[2 * e.pix], a.pixels[2 * e.pix + 1], e.sample,
                                     depth, org, dir, beta, vo);
            if (depth == 0u) {
                // a camera ray that leaves the scene has all-zero first-hit outputs: only z (0 = no hit, a hit has
                // z = t > 0) is stored for it, and accumulate_kernel reads the other nine planes for hits only
                if (h.hit) {
                    stp(S + 3 * P, pid, vo.normal.x);
                    stp(S + 4 * P, pid, vo.normal.y);
                    stp(S + 5 * P, pid, vo.normal.z);
                    stp(S + 6 * P, pid, vo.position.x);
                    stp(S + 7 * P, pid, vo.position.y);
                    stp(S + 8 * P, pid, vo.position.z);
                    stp(S + 9 * P, pid, vo.albedo.x);
                    stp(S + 10 * P, pid, vo.albedo.y);
                    stp(S + 11 * P, pid, vo.albedo.z);
                }
                stp(S + 12 * P, pid, vo.z);
                ray_depth = 0.0f + vo.t_add;
            } else if (h.hit && vo.t_add != 0.0f) {
                ray_depth = ray_depth + vo.t_add;
            }
            if (cont && depth + 1u == a.max_depth) {
                // the next vertex would be at the depth limit: it contributes beta * 0 (pathtracing.rs:24-26)
                vo.radiance = mul(beta, mk(0.0f, 0.0f, 0.0f));
                cont = false;
            }
            if (cont) {
                alive = true;
            } else if (depth == 0u && !h.hit) {
                // a camera ray that leaves the scene: its path is the sky colour and a ray_depth of 0, which
                // accumulate_kernel knows from z == 0 (stored above) without reading them
            } else {
                stp(S + 0 * P, pid, vo.radiance.x);
                stp(S + 1 * P, pid, vo.radiance.y);
                stp(S + 2 * P, pid, vo.radiance.z);
                stp(S + 13 * P, pid, ray_depth);
            }
        }
    }
    // ---- survivors keep their queue order inside the block; one atomic per block reserves the range
    const unsigned long long amask = __ballot(alive);
    if (lane == 0) wave_count[wave] = (uint32_t)__popcll(amask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < BLOCK / 64; ++w) total += wave_count[w];
        block_base = total ? atomicAdd(&a.counts[(size_t)((depth + 1u) * SEGMENTS + seg) * COUNTER_STRIDE], total) : 0u;
    }
    __syncthreads();
    if (alive) {
        uint32_t off = block_base;
        for (uint32_t w = 0; w < wave; ++w) off += wave_count[w];
        const uint32_t out = seg * a.seg_cap + off + (uint32_t)__popcll(amask & ((1ull << lane) - 1ull));
        stp(Qn + 0 * H, out, __builtin_bit_cast(float, pid));
        stp(Qn + 1 * H, out, org.x);
        stp(Qn + 2 * H, out, org.y);
        stp(Qn + 3 * H, out, org.z);
        stp(Qn + 4 * H, out, dir.x);
        stp(Qn + 5 * H, out, dir.y);
        stp(Qn + 6 * H, out, dir.z);
        stp(Qn + 7 * H, out, beta.x);
        stp(Qn + 8 * H, out, beta.y);
        stp(Qn + 9 * H, out, beta.z);
        stp(Qn + 10 * H, out, ray_depth);
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

// The thin tail of a pass: once few paths are left, a launch per bounce costs more than the bounce (the kernels are
// latency-bound with a few hundred thousand rays: the slowest ray of a wave takes ~150 dependent steps), so this kernel
// takes every path still alive at bounce `depth0` and runs it to its end, one path per lane: closest hit, shading, next
// ray, in the same per-ray functions as the kernels above (results do not depend on which kernel traces a path).  Waves
// are persistent and their lanes independent.  Traversal is extend_queue_kernel's: a lane at a leaf appends the record
// to the wave's queue and goes on, record phases take 64 entries whatever lanes they came from.  A lane whose traversal
// is over and whose entries have all been tested waits for a shading phase, which runs once RenderArgs::tail_shade
// lanes wait for it (shading costs several node steps) or nothing else can run; a lane whose path has ended takes the
// next entry of the queue of bounce depth0.  The launch lasts about as long as the deepest path while the other lanes
// work through the short ones.  After every shading phase the wave adds its surviving lanes to the counters of the
// bounces they enter, so the counters read like those of the wavefront kernels.
// Five waves per SIMD (96 registers, a dozen of them spilled around the shading step) measured faster than four without
// spills and than six with thirty.
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5))) void tail_kernel(RenderArgs a, uint32_t depth0) {
    __shared__ ExtendQueueLds lds;
    QueueStack st = make_stack_of<QUEUE_STACK>(lds.stack, a.spill);
    const size_t P = a.n_paths;
    const size_t H = (size_t)SEGMENTS * a.seg_cap;
    float *__restrict__ S = a.state;
    const float *__restrict__ Q = a.payload[depth0 & 1u];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave_base = __builtin_amdgcn_readfirstlane(tid & ~63u);
    WaveQueue wq{lds.queue[wave_base >> 6], 0u, 0u};
    QueueCursor cur;
    cur.begin(a, depth0);
    uint32_t last_post = 0;
    bool alive = false, exhausted = false;
    uint32_t pid = 0, depth = depth0; // pixel and sample are recomputed from the path id when the lane is shaded
    V3 beta = mk(0.0f, 0.0f, 0.0f);
    float ray_depth = 0.0f;
    Trav tr;
    tr.cur = RT_EMPTY;
    for (;;) {
        if (!exhausted && (int)__popcll(__ballot(alive)) < a.refill_below) {
            cur.claim(a, depth0);
            uint32_t i;
            if (cur.take(!alive, i)) {
                const uint32_t pos = cur.seg * a.seg_cap + i;
                const Entry e = queue_entry(a, depth0, cur.seg, i, Q + (size_t)cur.seg * a.seg_cap);
                pid = e.pid;
                const V3 org = mk(ldp(Q + 1 * H, pos), ldp(Q + 2 * H, pos), ldp(Q + 3 * H, pos));
                const V3 dir = mk(ldp(Q + 4 * H, pos), ldp(Q + 5 * H, pos), ldp(Q + 6 * H, pos));
                beta = mk(ldp(Q + 7 * H, pos), ldp(Q + 8 * H, pos), ldp(Q + 9 * H, pos));
                ray_depth = ldp(Q + 10 * H, pos);
                depth = depth0;
                queued_begin(lds, a.scene, tid, org, dir, tr, st);
                last_post = wq.head;
                alive = true;
            }
            exhausted = cur.finished();
        }
        if (__ballot(alive) == 0ull) {
            if (exhausted) break;
            continue;
        }
        st.note_depth();
        wq.post(alive, lane, tr, st, last_post);
        const bool at_leaf = alive && tr.cur != RT_EMPTY && (tr.cur & RT_LEAF) != 0u;
        const bool at_node = alive && tr.cur != RT_EMPTY && !at_leaf;
        const bool at_shade = alive && tr.cur == RT_EMPTY && !wq.outstanding(last_post);
        const uint32_t n_node = (uint32_t)__popcll(__ballot(at_node));
        const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(at_shade));
        // with no lane at a node or a leaf, the lanes that are not ready for shading wait for queued tests
        if (wq.length() > 0u && (wq.due(n_node, n_leaf, (uint32_t)a.queue_low) || (n_node == 0u && n_leaf == 0u && n_shade < (uint32_t)a.tail_shade))) {
            wq.test(lds, a.scene, wave_base, lane);
        } else if (n_shade > 0u && (n_shade >= (uint32_t)a.tail_shade || (n_node == 0u && n_leaf == 0u))) {
            if (at_shade) {
                float t;
                V3 ng, org, dir;
                uint32_t code;
                queued_result(lds, a.scene, tid, t, ng, code, org, dir);
                HitInfo h;
                h.u = h.v = 0.0f;
                h.prim_id = RT_EMPTY;
                h.hit = code != RT_EMPTY;
                h.t = t;
                h.pos = mk(0.0f, 0.0f, 0.0f);
                h.n = mk(0.0f, 0.0f, 0.0f);
                h.geom_id = RT_EMPTY;
                h.mat_id = 0;
                if (h.hit) {
                    h.pos = mk(org.x + t * dir.x, org.y + t * dir.y, org.z + t * dir.z);
                    h.n = normalize_or_zero(ng);
                    h.geom_id = code & RT_GEOM_MASK;
                    h.mat_id = h.geom_id < a.scene.n_geoms ? a.scene.geom_material[h.geom_id] : 0u;
                    if (h.mat_id >= a.scene.n_materials) h.mat_id = 0u;
                }
                const uint32_t per_block = a.samples * (uint32_t)PIXEL_BLOCK;
                const uint32_t blk = pid / per_block, rem = pid - blk * per_block;
                const uint32_t pix = blk * (uint32_t)PIXEL_BLOCK + rem % (uint32_t)PIXEL_BLOCK;
                const uint32_t sample = a.sample_base + rem / (uint32_t)PIXEL_BLOCK;
                VertexOut vo;
                bool cont = shade_vertex(a.scene, h, a.integrator, a.seed, a.pixels[2 * pix], a.pixels[2 * pix + 1], sample, depth, org, dir, beta, vo);
                if (h.hit && vo.t_add != 0.0f) ray_depth = ray_depth + vo.t_add;
                if (cont && depth + 1u == a.max_depth) {
                    // the next vertex would be at the depth limit: it contributes beta * 0 (pathtracing.rs:24-26)
                    vo.radiance = mul(beta, mk(0.0f, 0.0f, 0.0f));
                    cont = false;
                }
                if (!cont) {
                    stp(S + 0 * P, pid, vo.radiance.x);
                    stp(S + 1 * P, pid, vo.radiance.y);
                    stp(S + 2 * P, pid, vo.radiance.z);
                    stp(S + 13 * P, pid, ray_depth);
                    alive = false;
                } else {
                    ++depth;
                    queued_begin(lds, a.scene, tid, org, dir, tr, st);
                    last_post = wq.head;
                }
            }
            unsigned long long todo = __ballot(at_shade && alive);
            while (todo != 0ull) {
                const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)depth, (int)__builtin_ctzll(todo));
                const unsigned long long same = __ballot(at_shade && alive && depth == d);
                if (rank_in(~0ull) == 0u)
                    atomicAdd(&a.counts[(size_t)(d * SEGMENTS + cur.home) * COUNTER_STRIDE], (uint32_t)__popcll(same));
                todo &= ~same;
            }
        } else if (at_node) {
            queued_node_step(lds, a.scene, tid, tr, st);
        }
    }
}

// RaySeries::add_sample (crates/rt/src/renderer.rs:63-82, crates/rt/src/math/stat.rs:24-28,165-168) for the `samples`
// samples of each pixel, in sample order, filter weight 1 (crates/rt/src/filter.rs:30).
__global__ __launch_bounds__(BLOCK) void accumulate_kernel(RenderArgs a, float *__restrict__ accum, float count_inc) {
    const size_t P = a.n_paths;
    const size_t N = a.n_pixels;
    const float *__restrict__ S = a.state;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (size_t)gridDim.x * BLOCK) {
        if (a.pixel_done != nullptr && a.pixel_done[i] != 0u) continue; // stopped in an earlier pass of this call
        float acc[ACCUM_PLANES];
#pragma unroll
        for (int k = 0; k < ACCUM_PLANES; ++k) acc[k] = accum[(size_t)k * N + i];
        const size_t block_base = (i / PIXEL_BLOCK) * (size_t)a.samples * PIXEL_BLOCK + i % PIXEL_BLOCK;
        for (uint32_t s = 0; s < a.samples; ++s) {
            const size_t pid = block_base + (size_t)s * PIXEL_BLOCK;
            const float w = 1.0f;
            // z == 0: the camera ray hit nothing (a hit has z = t > 0).  Its radiance is the sky's under a throughput
            // of one (shade_vertex: beta * sky, integrators/mod.rs:19), its ray_depth 0, its first-hit outputs zeros;
            // none of them was stored
            const float z = S[12 * P + pid];
            const bool sky = z == 0.0f;
            const float col[3] = {sky ? 1.0f * 0.5f : S[0 * P + pid], sky ? 1.0f * 0.3f : S[1 * P + pid],
                                  sky ? 1.0f * 1.0f : S[2 * P + pid]};
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                acc[ch] += col[ch];
                acc[3 + ch] += col[ch] * col[ch];
                acc[6 + ch] = acc[6 + ch] + w * col[ch];
            }
            acc[9] += w;
            // the sums never hold -0, so leaving out the addition of the +0 first-hit outputs of such a ray changes no bit
            if (!sky) {
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[10 + k] = acc[10 + k] + S[(size_t)(3 + k) * P + pid];
            }
            acc[19] += z;
            acc[20] += sky ? 0.0f : S[13 * P + pid];
            acc[21] += count_inc;
            // crates/rt-cli/src/executor.rs:244-249: the pixel is done for this batch once all channels are precise
            if (a.allowed_error > 0.0f && channel_precise(acc[21], acc[0], acc[3], a.allowed_error) &&
                channel_precise(acc[21], acc[1], acc[4], a.allowed_error) &&
                channel_precise(acc[21], acc[2], acc[5], a.allowed_error)) {
                if (a.pixel_done != nullptr) {
                    a.pixel_done[i] = 1u;
                    atomicAdd(&a.pixel_done[N], 1u);
                }
                break;
            }
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

// Position of every listed pixel in a row-major frame window; pixels outside the window raise *bad.
__global__ __launch_bounds__(BLOCK) void frame_index_kernel(const uint32_t *__restrict__ xy, uint32_t n, uint32_t ox,
                                                            uint32_t oy, uint32_t w, uint32_t h,
                                                            uint32_t *__restrict__ lin, uint32_t *bad) {
    const uint32_t i = blockIdx.x * (uint32_t)BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = xy[2 * i] - ox, y = xy[2 * i + 1] - oy; // wraps to a large value when left of / above the window
    if (x >= w || y >= h) {
        lin[i] = 0u;
        atomicOr(bad, 1u);
    } else {
        lin[i] = y * w + x;
    }
}

// dst[k][lin[i]] = src[k][i] for the ACCUM_PLANES planes of n source pixels (plane strides n and n_dst).
__global__ __launch_bounds__(BLOCK) void scatter_planes_kernel(const float *__restrict__ src, uint32_t n,
                                                               const uint32_t *__restrict__ lin,
                                                               float *__restrict__ dst, uint32_t n_dst) {
    const uint32_t i = blockIdx.x * (uint32_t)BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = lin[i];
#pragma unroll
    for (int k = 0; k < ACCUM_PLANES; ++k) dst[(size_t)k * n_dst + j] = src[(size_t)k * n + i];
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

int launch_extend(const RenderArgs &args, uint32_t depth, uint32_t grid_blocks, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (args.queue_low > 0 && args.visit_counters == nullptr) // visits are counted on the undeferred traversal
        hipLaunchKernelGGL(extend_queue_kernel, dim3(grid_blocks), dim3(BLOCK), 0, s, args, depth);
    else
        hipLaunchKernelGGL(extend_kernel, dim3(grid_blocks), dim3(BLOCK), 0, s, args, depth);
    return check_launch();
}

int launch_shade(const RenderArgs &args, uint32_t depth, uint32_t max_segment_entries, void *stream) {
    if (max_segment_entries == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t per_seg = (max_segment_entries + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(shade_kernel, dim3(per_seg * SEGMENTS), dim3(BLOCK), 0, s, args, depth);
    return check_launch();
}

int launch_tail(const RenderArgs &args, uint32_t depth0, uint32_t grid_blocks, void *stream) {
    if (grid_blocks == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tail_kernel, dim3(grid_blocks), dim3(BLOCK), 0, s, args, depth0);
    return check_launch();
}

int tail_blocks_per_cu() {
    int n = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, tail_kernel, BLOCK, 0) == hipSuccess && n >= 1 ? n : 1;
}

int launch_raygen(const RenderArgs &args, void *stream) {
    if (args.seg_cap == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t per_seg = (args.seg_cap + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(raygen_kernel, dim3(per_seg * SEGMENTS), dim3(BLOCK), 0, s, args);
    return check_launch();
}

int extend_blocks_per_cu(bool queued) {
    int n = 0;
    const hipError_t e = queued ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, extend_queue_kernel, BLOCK, 0)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, extend_kernel, BLOCK, 0);
    return e == hipSuccess && n >= 1 ? n : 1;
}

int launch_accumulate(const RenderArgs &args, float *d_accum, float count_inc, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const uint32_t grid = blocks_for(args.n_pixels, 256u * 16u);
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid), dim3(BLOCK), 0, s, args, d_accum, count_inc);
    return check_launch();
}

int launch_frame_index(const uint32_t *d_xy, uint32_t n, uint32_t ox, uint32_t oy, uint32_t w, uint32_t h,
                       uint32_t *d_lin, uint32_t *d_bad, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(frame_index_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, d_xy, n, ox,
                       oy, w, h, d_lin, d_bad);
    return check_launch();
}

int launch_scatter_planes(const float *d_src, uint32_t n, const uint32_t *d_lin, float *d_dst, uint32_t n_dst,
                          void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(scatter_planes_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, d_src,
                       n, d_lin, d_dst, n_dst);
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
