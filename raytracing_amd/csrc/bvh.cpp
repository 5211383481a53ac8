// bvh.cpp -- binned-SAH BVH2 build with spatial splits (multi-threaded over subtrees), collapse to BVH4 nodes of 64
// bytes with 8-bit child boxes, placement of nodes and record slots in one array of 64-byte items.
#include "bvh.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

namespace rtb {

namespace {

constexpr int BINS = 16;
constexpr uint32_t LEAF_MAX = 1; // items per leaf; the scene builder passes primitive records of up to two triangles; measured on a
                                 // 5 M-triangle mesh, one record per leaf is 6 % faster than two
constexpr uint32_t PARALLEL_MIN = 32768; // subtrees at least this large may get their own thread
constexpr uint32_t MAX_CUTS = 32;         // spatial splits on one path from the root (bounds the depth they can add)

struct Node2 {
    Box box;
    uint32_t left = 0, right = 0; // children (inner)
    uint32_t first = 0, count = 0; // leaf range in order[] (count > 0 = leaf)
};

inline void box_init(Box &b) {
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = INFINITY;
        b.hi[a] = -INFINITY;
    }
}
inline void box_grow(Box &b, const Box &o) {
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(b.lo[a], o.lo[a]);
        b.hi[a] = fmaxf(b.hi[a], o.hi[a]);
    }
}
inline float half_area(const Box &b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}

struct Ref {
    Box box;       // padded bounds of the part of the primitive this reference stands for
    uint32_t prim; // index of the input primitive
};

inline float centre_of(const Box &b, int a) { return 0.5f * b.lo[a] + 0.5f * b.hi[a]; }
inline Box box_meet(const Box &x, const Box &y) {
    Box r;
    for (int a = 0; a < 3; ++a) {
        r.lo[a] = fmaxf(x.lo[a], y.lo[a]);
        r.hi[a] = fminf(x.hi[a], y.hi[a]);
    }
    return r;
}
inline bool ref_before(const Ref &x, const Ref &y, int a) {
    const float cx = centre_of(x.box, a), cy = centre_of(y.box, a);
    return cx < cy || (cx == cy && x.prim < y.prim);
}

// Binary SAH build over primitive references.  A subtree owns refs[begin, cap): its references in [begin, end) and
// free room behind them, which spatial splits fill with the second halves of the references they cut; and the node
// slots [slot, slot + 2 * (cap - begin) - 1).  Object splits come from 16 bins per axis (a full sweep for a handful of
// references); where the two sides of the best object split overlap and a reference is large against the node, the
// best of 16 spatial planes per axis (references chopped into the bins by clipping the primitive itself) is taken
// instead when it is cheaper (Stich et al., "Spatial Splits in Bounding Volume Hierarchies", 2009).  With no clip
// function, or where no spatial split pays, the tree is the object-split tree.
struct Builder {
    std::vector<Ref> refs;
    std::vector<uint8_t> used; // refs[i] is a leaf of the finished tree
    std::vector<Node2> nodes;
    const ClipFn *clip = nullptr;
    float root_area = 0.0f;
    float alpha = 1e-5f;       // spatial candidates only where the object split's sides overlap by more than this share of the root's area
    float large = 0.125f;      // ... and some reference's box has at least this share of the node's area (a plane through
                               // references that are all small against the node moves its children's boxes by little)
    float gain = 0.98f;        // ... and taken only when at most this share of the object split's cost
    std::atomic<int> live{1};
    int max_threads = 1;
    std::atomic<size_t> done{0};
    std::atomic<size_t> spatial_splits{0};
    std::atomic<bool> cancelled{false};

    uint32_t sweep_max = 16; // subtrees of at most this many items are split by a full sweep: bins are too coarse for a
                             // handful of items (a 37-primitive scene: 14 % fewer node visits per ray); on large meshes
                             // it changes nothing measurable, for any limit
    // per-thread scratch
    static std::vector<float> &sweep_area(uint32_t n) {
        thread_local std::vector<float> v;
        if (v.size() < n + 1) v.resize(n + 1);
        return v;
    }
    static std::vector<Ref> &side(int which) {
        thread_local std::vector<Ref> v[2];
        return v[which];
    }

    bool clipped(const Ref &r, const Box &region, Box &out) const {
        if (!(*clip)(r.prim, region, out)) return false;
        out = box_meet(out, r.box); // a part never sticks out of the reference it was cut from
        for (int a = 0; a < 3; ++a)
            if (!(out.lo[a] <= out.hi[a])) return false;
        return true;
    }

    // Best of SPATIAL_BINS - 1 planes per axis; false when nothing was found.
    bool spatial_candidate(uint32_t begin, uint32_t end, const Box &bb, int &axis, float &pos, float &cost) const {
        cost = INFINITY;
        axis = -1;
        for (int a = 0; a < 3; ++a) {
            const float ext = bb.hi[a] - bb.lo[a];
            if (!(ext > 0.0f) || !isfinite(ext)) continue;
            const float width = ext / (float)BINS, inv = (float)BINS / ext;
            Box bins[BINS];
            uint32_t enter[BINS], leave[BINS];
            for (int b = 0; b < BINS; ++b) {
                box_init(bins[b]);
                enter[b] = leave[b] = 0;
            }
            auto bin_of = [&](float x) {
                const int b = (int)((x - bb.lo[a]) * inv);
                return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
            };
            for (uint32_t i = begin; i < end; ++i) {
                const Ref &r = refs[i];
                const int b0 = bin_of(r.box.lo[a]), b1 = bin_of(r.box.hi[a]);
                if (b0 == b1) {
                    box_grow(bins[b0], r.box);
                } else {
                    for (int b = b0; b <= b1; ++b) {
                        Box region = r.box, part;
                        region.lo[a] = fmaxf(r.box.lo[a], bb.lo[a] + (float)b * width);
                        region.hi[a] = fminf(r.box.hi[a], bb.lo[a] + (float)(b + 1) * width);
                        if (clipped(r, region, part)) box_grow(bins[b], part);
                    }
                }
                ++enter[b0];
                ++leave[b1];
            }
            float rarea[BINS];
            uint32_t rcnt[BINS];
            Box r;
            box_init(r);
            uint32_t rc = 0;
            for (int b = BINS - 1; b > 0; --b) {
                box_grow(r, bins[b]);
                rc += leave[b];
                rarea[b] = half_area(r);
                rcnt[b] = rc;
            }
            Box l;
            box_init(l);
            uint32_t lc = 0;
            for (int b = 0; b < BINS - 1; ++b) {
                box_grow(l, bins[b]);
                lc += enter[b];
                if (lc == 0 || rcnt[b + 1] == 0) continue;
                const float c = half_area(l) * (float)lc + rarea[b + 1] * (float)rcnt[b + 1];
                if (c < cost) {
                    cost = c;
                    axis = a;
                    pos = bb.lo[a] + (float)(b + 1) * width;
                }
            }
        }
        return axis >= 0;
    }

    // Cuts refs[begin, end) at the plane into the two per-thread side vectors; false when the cut is useless or its
    // halves do not fit the subtree's room.
    bool spatial_cut(uint32_t begin, uint32_t end, uint32_t cap, int a, float pos) const {
        std::vector<Ref> &L = side(0), &R = side(1);
        L.clear();
        R.clear();
        const uint32_t count = end - begin;
        for (uint32_t i = begin; i < end; ++i) {
            const Ref &r = refs[i];
            if (r.box.hi[a] <= pos) {
                L.push_back(r);
            } else if (r.box.lo[a] >= pos) {
                R.push_back(r);
            } else {
                Box lr = r.box, rr = r.box, lp, rp;
                lr.hi[a] = pos;
                rr.lo[a] = pos;
                const bool hl = clipped(r, lr, lp), hr = clipped(r, rr, rp);
                if (hl && hr) {
                    L.push_back(Ref{lp, r.prim});
                    R.push_back(Ref{rp, r.prim});
                } else if (hr) {
                    R.push_back(r);
                } else {
                    L.push_back(r);
                }
            }
            if (L.size() + R.size() + (end - i - 1) > (size_t)(cap - begin)) return false;
        }
        if (L.empty() || R.empty()) return false;
        if (L.size() == count && R.size() == count) return false; // every reference on both sides: nothing separated
        return true;
    }

    // Subtree over refs[begin, end) with room up to cap, rooted at node slot `slot`, below `cuts` spatial splits.
    void build(uint32_t slot, uint32_t begin, uint32_t end, uint32_t cap, uint32_t cuts) {
        Node2 &n = nodes[slot];
        Box bb, cb;
        box_init(bb);
        box_init(cb);
        float max_ref_area = 0.0f;
        for (uint32_t i = begin; i < end; ++i) {
            const Box &rb = refs[i].box;
            box_grow(bb, rb);
            max_ref_area = fmaxf(max_ref_area, half_area(rb));
            for (int a = 0; a < 3; ++a) {
                const float c = centre_of(rb, a);
                cb.lo[a] = fminf(cb.lo[a], c);
                cb.hi[a] = fmaxf(cb.hi[a], c);
            }
        }
        n.box = bb;
        const uint32_t count = end - begin;
        if (count <= LEAF_MAX) {
            n.first = begin;
            n.count = count;
            for (uint32_t i = begin; i < end; ++i) used[i] = 1;
            done.fetch_add(count, std::memory_order_relaxed);
            return;
        }
        // ---- best object split: obj_axis < 0 = none (all centres equal), then the range is halved
        float obj_cost = INFINITY;
        int obj_axis = -1, obj_bin = -1;
        uint32_t obj_left = 0;
        Box obj_l = bb, obj_r = bb;
        float scale[3] = {0.0f, 0.0f, 0.0f};
        auto bin_of = [&](const Ref &r, int a) {
            int b = (int)((centre_of(r.box, a) - cb.lo[a]) * scale[a]);
            return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
        };
        const bool sweep = count <= sweep_max;
        if (sweep) {
            // few items: every split position on every axis (items sorted by centre), instead of bin boundaries
            std::vector<Ref> &tmp = side(0);
            std::vector<float> &rarea = sweep_area(count);
            for (int a = 0; a < 3; ++a) {
                tmp.assign(refs.begin() + begin, refs.begin() + end);
                std::sort(tmp.begin(), tmp.end(), [&](const Ref &x, const Ref &y) { return ref_before(x, y, a); });
                Box r;
                box_init(r);
                for (uint32_t i = count; i-- > 1;) {
                    box_grow(r, tmp[i].box);
                    rarea[i] = half_area(r);
                }
                Box l;
                box_init(l);
                for (uint32_t i = 1; i < count; ++i) {
                    box_grow(l, tmp[i - 1].box);
                    const float cost = half_area(l) * (float)i + rarea[i] * (float)(count - i);
                    if (cost < obj_cost) {
                        obj_cost = cost;
                        obj_axis = a;
                        obj_left = i;
                    }
                }
            }
        } else {
            // bin on all three axes, keep the cheapest split
            Box bins[3][BINS];
            uint32_t bcnt[3][BINS];
            for (int a = 0; a < 3; ++a) {
                const float ext = cb.hi[a] - cb.lo[a];
                scale[a] = ext > 0.0f ? (float)BINS / ext : 0.0f;
                for (int b = 0; b < BINS; ++b) {
                    box_init(bins[a][b]);
                    bcnt[a][b] = 0;
                }
            }
            for (uint32_t i = begin; i < end; ++i) {
                const Ref &r = refs[i];
                for (int a = 0; a < 3; ++a) {
                    if (scale[a] == 0.0f) continue;
                    const int b = bin_of(r, a);
                    bcnt[a][b]++;
                    box_grow(bins[a][b], r.box);
                }
            }
            for (int a = 0; a < 3; ++a) {
                if (scale[a] == 0.0f) continue;
                float rarea[BINS];
                uint32_t rcnt[BINS];
                Box r;
                box_init(r);
                uint32_t rc = 0;
                for (int b = BINS - 1; b > 0; --b) {
                    box_grow(r, bins[a][b]);
                    rc += bcnt[a][b];
                    rarea[b] = half_area(r);
                    rcnt[b] = rc;
                }
                Box l;
                box_init(l);
                uint32_t lc = 0;
                for (int b = 0; b < BINS - 1; ++b) {
                    box_grow(l, bins[a][b]);
                    lc += bcnt[a][b];
                    if (lc == 0 || rcnt[b + 1] == 0) continue;
                    const float cost = half_area(l) * (float)lc + rarea[b + 1] * (float)rcnt[b + 1];
                    if (cost < obj_cost) {
                        obj_cost = cost;
                        obj_axis = a;
                        obj_bin = b;
                    }
                }
            }
        }
        // ---- put the object split in place (the spatial candidate needs its two boxes)
        uint32_t mid = begin + count / 2;
        if (sweep) {
            const int a = obj_axis < 0 ? 0 : obj_axis;
            std::sort(refs.begin() + begin, refs.begin() + end, [&](const Ref &x, const Ref &y) { return ref_before(x, y, a); });
            if (obj_axis >= 0) mid = begin + obj_left;
        } else if (obj_axis >= 0) {
            uint32_t i = begin, j = end;
            while (i < j) {
                if (bin_of(refs[i], obj_axis) <= obj_bin) {
                    ++i;
                } else {
                    --j;
                    std::swap(refs[i], refs[j]);
                }
            }
            if (i > begin && i < end) mid = i;
        }
        uint32_t nl = mid - begin, nr = end - mid;
        // ---- spatial split instead?
        if (clip != nullptr && cap > end && cuts < MAX_CUTS && root_area > 0.0f && max_ref_area >= large * half_area(bb)) {
            box_init(obj_l);
            box_init(obj_r);
            for (uint32_t i = begin; i < mid; ++i) box_grow(obj_l, refs[i].box);
            for (uint32_t i = mid; i < end; ++i) box_grow(obj_r, refs[i].box);
            const float split_cost = half_area(obj_l) * (float)nl + half_area(obj_r) * (float)nr;
            if (half_area(box_meet(obj_l, obj_r)) > alpha * root_area) {
                int sa = -1;
                float sp = 0.0f, sc = INFINITY;
                if (spatial_candidate(begin, end, bb, sa, sp, sc) && sc < gain * split_cost && spatial_cut(begin, end, cap, sa, sp)) {
                    const std::vector<Ref> &L = side(0), &R = side(1);
                    nl = (uint32_t)L.size();
                    nr = (uint32_t)R.size();
                    std::copy(L.begin(), L.end(), refs.begin() + begin);
                    std::copy(R.begin(), R.end(), refs.begin() + begin + nl);
                    mid = begin + nl;
                    end = mid + nr;
                    spatial_splits.fetch_add(1, std::memory_order_relaxed);
                    ++cuts;
                }
            }
        }
        // ---- the free room is shared out by size; the right side moves up behind the left side's room
        const uint32_t room = cap - end;
        const uint32_t room_l = (uint32_t)((uint64_t)room * nl / (nl + nr));
        if (room_l > 0) memmove(&refs[mid + room_l], &refs[mid], (size_t)nr * sizeof(Ref));
        const uint32_t l_cap = mid + room_l, r_begin = l_cap, r_end = r_begin + nr;
        const uint32_t ls = slot + 1, rs = slot + 2 * (l_cap - begin);
        n.left = ls;
        n.right = rs;
        n.count = 0;
        if (cancelled.load(std::memory_order_relaxed)) return;
        bool spawned = false;
        std::thread worker;
        if (nl >= PARALLEL_MIN && nr >= PARALLEL_MIN) {
            int cur = live.load();
            while (cur < max_threads && !live.compare_exchange_weak(cur, cur + 1)) {
            }
            if (cur < max_threads) {
                spawned = true;
                worker = std::thread([this, ls, begin, mid, l_cap, cuts] {
                    build(ls, begin, mid, l_cap, cuts);
                    live.fetch_sub(1);
                });
            }
        }
        if (!spawned) build(ls, begin, mid, l_cap, cuts);
        build(rs, r_begin, r_end, cap, cuts);
        if (spawned) worker.join();
    }
};

} // namespace

bool build(const std::vector<Box> &boxes, WideBvh &out, const std::function<bool(double)> &progress, unsigned threads,
           const ClipFn &clip) {
    out = WideBvh();
    const size_t n = boxes.size();
    if (progress && !progress(0.0)) return false;
    if (n == 0) {
        if (progress) progress(1.0);
        return true;
    }
    Builder b;
    bool spatial = (bool)clip;
    if (const char *v = getenv("RTMI_SPATIAL")) spatial = spatial && atoi(v) != 0;
    if (const char *v = getenv("RTMI_SPATIAL_ALPHA")) b.alpha = (float)atof(v);
    if (const char *v = getenv("RTMI_SPATIAL_LARGE")) b.large = (float)atof(v);
    if (const char *v = getenv("RTMI_SPATIAL_GAIN")) b.gain = (float)atof(v); // above 1: cuts that cost more are taken too (tests)
    // room for the references spatial splits add: as many again as there are primitives (Embree's split factor for
    // its high-quality build is 2 as well)
    size_t room = spatial ? n + 256 : 0;
    if (const char *v = getenv("RTMI_SPATIAL_ROOM")) room = spatial ? (size_t)(atof(v) * (double)n) + 256 : 0;
    const size_t cap = n + room;
    b.clip = spatial ? &clip : nullptr;
    b.refs.resize(cap);
    b.used.assign(cap, 0);
    b.nodes.resize(2 * cap);
    Box all;
    box_init(all);
    for (size_t i = 0; i < n; ++i) {
        b.refs[i].box = boxes[i];
        b.refs[i].prim = (uint32_t)i;
        box_grow(all, boxes[i]);
    }
    b.root_area = half_area(all);
    unsigned hw = threads ? threads : std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (hw > 16) hw = 16;
    b.max_threads = (int)hw;
    if (const char *v = getenv("RTMI_SWEEP_MAX")) b.sweep_max = (uint32_t)atoi(v);

    // run the build on a worker so that this thread can report progress
    std::atomic<bool> finished{false};
    std::thread root([&] {
        b.build(0, 0, (uint32_t)n, (uint32_t)cap, 0);
        finished.store(true);
    });
    bool ok = true;
    while (!finished.load()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(n > 100000 ? 20 : 1));
        if (progress && ok && !progress(std::min(0.8, 0.8 * (double)b.done.load() / (double)n))) {
            ok = false;
            b.cancelled.store(true);
        }
    }
    root.join();
    if (!ok) return false;

    // leaf slots: the references that became leaves, in the order they lie in
    std::vector<uint32_t> slot_of(cap, rtd::RT_EMPTY);
    for (size_t i = 0; i < cap; ++i)
        if (b.used[i]) {
            slot_of[i] = (uint32_t)out.prim_order.size();
            out.prim_order.push_back(b.refs[i].prim);
        }
    out.spatial_splits = (uint32_t)b.spatial_splits.load();

    // collapse to 4-wide, breadth-first
    const std::vector<Node2> &N2 = b.nodes;
    std::vector<uint32_t> wide_root; // node2 index each wide node expands
    wide_root.push_back(0);
    std::vector<rtd::F4> W; // the nodes, breadth-first, child references by breadth-first index / leaf slot
    std::vector<uint8_t> nchild;
    std::vector<uint32_t> child_ref; // 4 per wide node
    const float root_area = half_area(N2[0].box);
    double sah = 0.0;
    for (size_t w = 0; w < wide_root.size(); ++w) {
        const Node2 &r = N2[wide_root[w]];
        uint32_t ch[4];
        int nc = 0;
        if (r.count > 0) {
            ch[nc++] = wide_root[w]; // a root that is itself a leaf
        } else {
            ch[nc++] = r.left;
            ch[nc++] = r.right;
            while (nc < 4) {
                int pick = -1;
                float best_area = -1.0f;
                for (int k = 0; k < nc; ++k) {
                    if (N2[ch[k]].count > 0) continue;
                    const float ar = half_area(N2[ch[k]].box);
                    if (ar > best_area) {
                        best_area = ar;
                        pick = k;
                    }
                }
                if (pick < 0) break;
                const uint32_t c = ch[pick];
                for (int k = nc; k > pick + 1; --k) ch[k] = ch[k - 1];
                ch[pick] = N2[c].left;
                ch[pick + 1] = N2[c].right;
                ++nc;
            }
        }
        // Quantisation frame of this node: origin = low corner of the union of its child boxes, one power-of-two step
        // per axis such that 255 steps span the union.  Child planes are rounded outwards, so a decoded box always
        // contains the box it stands for.
        Box u;
        box_init(u);
        for (int k = 0; k < nc; ++k) box_grow(u, N2[ch[k]].box);
        float org[3], step[3];
        for (int a = 0; a < 3; ++a) {
            org[a] = isfinite(u.lo[a]) ? u.lo[a] : 0.0f;
            const double ext = (double)u.hi[a] - (double)org[a];
            int e = -100;
            if (!(ext <= 1e14)) // non-finite or huge: such boxes are not representable (and not supported)
                e = 40;
            else if (ext > 0.0) {
                e = (int)ceil(log2(ext / 255.0));
                if (ldexp(255.0, e) < ext) ++e;
                if (e < -100) e = -100;
            }
            step[a] = (float)ldexp(1.0, e);
        }
        uint32_t qw[6] = {0, 0, 0, 0, 0, 0}; // lo.x, lo.y, lo.z, hi.x, hi.y, hi.z; child k in byte k
        uint32_t refs[4];
        for (int k = 0; k < 4; ++k) {
            if (k < nc) {
                const Node2 &c = N2[ch[k]];
                for (int a = 0; a < 3; ++a) {
                    const double o = org[a], st = step[a];
                    double ql = floor(((double)c.box.lo[a] - o) / st), qh = ceil(((double)c.box.hi[a] - o) / st);
                    if (!(ql >= 0.0)) ql = 0.0;
                    if (!(qh <= 255.0)) qh = 255.0;
                    if (!(ql <= 255.0)) ql = 255.0;
                    if (!(qh >= 0.0)) qh = 0.0;
                    while (ql > 0.0 && o + ql * st > (double)c.box.lo[a]) ql -= 1.0;
                    while (qh < 255.0 && o + qh * st < (double)c.box.hi[a]) qh += 1.0;
                    qw[a] |= (uint32_t)ql << (8 * k);
                    qw[3 + a] |= (uint32_t)qh << (8 * k);
                }
                if (c.count > 0) {
                    refs[k] = rtd::RT_LEAF | ((c.count - 1u) << 28) | slot_of[c.first];
                    if (root_area > 0.0f) sah += (double)half_area(c.box) / root_area * c.count;
                } else {
                    refs[k] = (uint32_t)wide_root.size();
                    wide_root.push_back(ch[k]);
                    if (root_area > 0.0f) sah += (double)half_area(c.box) / root_area;
                }
            } else {
                refs[k] = rtd::RT_EMPTY;
            }
        }
        rtd::F4 slots[rtd::RT_NODE_F4];
        float *f = reinterpret_cast<float *>(slots);
        f[0] = org[0];
        f[1] = org[1];
        f[2] = org[2];
        f[3] = step[0];
        f[4] = step[1];
        f[5] = step[2];
        memcpy(&f[6], qw, sizeof qw);
        memcpy(&f[12], refs, sizeof refs);
        for (int k = 0; k < rtd::RT_NODE_F4; ++k) W.push_back(slots[k]);
        nchild.push_back((uint8_t)nc);
        for (int k = 0; k < 4; ++k) child_ref.push_back(refs[k]);
        if (progress && (w & 0xFFFF) == 0 && !progress(0.8 + 0.1 * (double)w / (double)(wide_root.size() + 1)))
            return false;
    }
    out.n_nodes = (uint32_t)wide_root.size();
    out.sah_cost = (float)sah;

    // worst-case stack need and depth: children always have larger indices than their parent
    std::vector<uint32_t> need(out.n_nodes, 0), depth(out.n_nodes, 1);
    for (size_t w = out.n_nodes; w-- > 0;) {
        uint32_t nd = 0, dp = 1;
        const uint32_t nc = nchild[w];
        for (uint32_t k = 0; k < nc; ++k) {
            const uint32_t ref = child_ref[4 * w + k];
            uint32_t cn = 0, cd = 0;
            if (!(ref & rtd::RT_LEAF)) {
                cn = need[ref];
                cd = depth[ref];
            }
            nd = std::max(nd, (nc - 1) + cn);
            dp = std::max(dp, 1 + cd);
        }
        need[w] = nd;
        depth[w] = dp;
    }
    out.stack_need = need[0];
    out.depth = depth[0];

    // ---- placement: the children of a node, inner nodes and records alike, take consecutive items in slot order, so
    // that the node names them by one index (and siblings, which a ray that reaches the node tends to visit together,
    // share lines).  Groups go breadth-first and start on a line boundary; the root has the first line to itself.
    static_assert(LEAF_MAX == 1, "a leaf is one record: child k of a node is item first + k");
    const uint32_t n_nodes = out.n_nodes, n_slots = (uint32_t)out.prim_order.size();
    std::vector<uint32_t> node_item(n_nodes, rtd::RT_EMPTY), rec_item(n_slots, rtd::RT_EMPTY), first_item(n_nodes, 0);
    uint32_t next = 2;
    if (n_nodes > 0) node_item[0] = 0;
    for (uint32_t w = 0; w < n_nodes; ++w) { // a parent has a smaller breadth-first index than its children
        const uint32_t nc = nchild[w];
        first_item[w] = next;
        for (uint32_t k = 0; k < nc; ++k) {
            const uint32_t c = child_ref[4 * (size_t)w + k];
            if (c & rtd::RT_LEAF)
                rec_item[c & 0x0FFFFFFFu] = next + k;
            else
                node_item[c] = next + k;
        }
        next = (next + nc + 1u) & ~1u;
    }
    out.n_items = next;
    out.items.assign((size_t)next * rtd::RT_NODE_F4, rtd::F4{0.0f, 0.0f, 0.0f, 0.0f});
    for (uint32_t w = 0; w < n_nodes; ++w) {
        // W[w] holds origin, steps, planes and the breadth-first child refs; the item gets the packed form the kernels
        // read (rt_device.h) in its first 48 bytes and the children's item refs, one word each, in the last 16
        const float *src = reinterpret_cast<const float *>(&W[(size_t)w * rtd::RT_NODE_F4]);
        uint32_t word[16];
        memcpy(word, src, 64);
        uint32_t step_hi[3];
        for (int a = 0; a < 3; ++a) step_hi[a] = word[3 + a] >> 16; // a power of two: the low 16 bits are zero
        uint32_t leaf_bits = 0, refs[4];
        const uint32_t nc = nchild[w];
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t c = child_ref[4 * (size_t)w + k];
            if (c == rtd::RT_EMPTY) {
                refs[k] = c;
            } else if (c & rtd::RT_LEAF) {
                leaf_bits |= 1u << k;
                refs[k] = rtd::RT_LEAF | (first_item[w] + k);
            } else {
                refs[k] = first_item[w] + k;
            }
        }
        uint32_t item[16];
        memcpy(&item[0], &word[0], 12);                 // origin
        item[3] = step_hi[0] | (step_hi[1] << 16);      // steps x, y
        memcpy(&item[4], &word[6], 24);                 // planes
        item[10] = step_hi[2] | (leaf_bits << 16) | ((nc - 1u) << 24);
        item[11] = first_item[w];
        memcpy(&item[12], refs, sizeof refs);
        memcpy(&out.items[(size_t)node_item[w] * rtd::RT_NODE_F4], item, 64);
    }
    out.record_item = std::move(rec_item);
    if (progress && !progress(0.9)) return false;
    return true;
}

} // namespace rtb
