// bvh.cpp -- binned-SAH BVH2 build (multi-threaded over subtrees), collapse to BVH4 nodes of 64 bytes with 8-bit child
// boxes, placement of nodes and record slots in one array of 64-byte items.
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

struct Builder {
    const std::vector<Box> &boxes;
    std::vector<float> cen; // 3 per primitive
    std::vector<uint32_t> order;
    std::vector<Node2> nodes;
    std::atomic<int> live{1};
    int max_threads = 1;
    std::atomic<size_t> done{0};
    std::atomic<bool> cancelled{false};

    uint32_t sweep_max = 16; // subtrees of at most this many items are split by a full sweep: bins are too coarse for a
                             // handful of items (a 37-primitive scene: 14 % fewer node visits per ray); on large meshes
                             // it changes nothing measurable, for any limit
    explicit Builder(const std::vector<Box> &b) : boxes(b) {}
    // per-thread scratch of the sweep
    static std::vector<uint32_t> &sweep_tmp(uint32_t n) {
        thread_local std::vector<uint32_t> v;
        if (v.size() < n) v.resize(n);
        return v;
    }
    static std::vector<float> &sweep_area(uint32_t n) {
        thread_local std::vector<float> v;
        if (v.size() < n + 1) v.resize(n + 1);
        return v;
    }

    // Subtree over order[begin, end) rooted at node slot `slot`; it owns slots [slot, slot + 2 * (end - begin) - 1).
    void build(uint32_t slot, uint32_t begin, uint32_t end) {
        Node2 &n = nodes[slot];
        Box bb, cb;
        box_init(bb);
        box_init(cb);
        for (uint32_t i = begin; i < end; ++i) {
            const uint32_t p = order[i];
            box_grow(bb, boxes[p]);
            for (int a = 0; a < 3; ++a) {
                cb.lo[a] = fminf(cb.lo[a], cen[3 * (size_t)p + a]);
                cb.hi[a] = fmaxf(cb.hi[a], cen[3 * (size_t)p + a]);
            }
        }
        n.box = bb;
        const uint32_t count = end - begin;
        if (count <= LEAF_MAX) {
            n.first = begin;
            n.count = count;
            done.fetch_add(count, std::memory_order_relaxed);
            return;
        }
        if (count <= sweep_max) {
            // few items: every split position on every axis (items sorted by centre), instead of bin boundaries
            std::vector<uint32_t> &tmp = sweep_tmp(count);
            std::vector<float> &rarea = sweep_area(count);
            float best = INFINITY;
            int best_axis = -1;
            uint32_t best_left = 0;
            for (int a = 0; a < 3; ++a) {
                std::copy(order.begin() + begin, order.begin() + end, tmp.begin());
                std::sort(tmp.begin(), tmp.begin() + count, [&](uint32_t x, uint32_t y) {
                    const float cx = cen[3 * (size_t)x + a], cy = cen[3 * (size_t)y + a];
                    return cx < cy || (cx == cy && x < y);
                });
                Box r;
                box_init(r);
                for (uint32_t i = count; i-- > 1;) {
                    box_grow(r, boxes[tmp[i]]);
                    rarea[i] = half_area(r);
                }
                Box l;
                box_init(l);
                for (uint32_t i = 1; i < count; ++i) {
                    box_grow(l, boxes[tmp[i - 1]]);
                    const float cost = half_area(l) * (float)i + rarea[i] * (float)(count - i);
                    if (cost < best) {
                        best = cost;
                        best_axis = a;
                        best_left = i;
                    }
                }
            }
            const int a = best_axis < 0 ? 0 : best_axis;
            std::sort(order.begin() + begin, order.begin() + end, [&](uint32_t x, uint32_t y) {
                const float cx = cen[3 * (size_t)x + a], cy = cen[3 * (size_t)y + a];
                return cx < cy || (cx == cy && x < y);
            });
            const uint32_t nl = best_axis < 0 ? count / 2 : best_left;
            const uint32_t mid = begin + nl;
            const uint32_t ls = slot + 1, rs = slot + 2 * nl;
            n.left = ls;
            n.right = rs;
            n.count = 0;
            if (cancelled.load(std::memory_order_relaxed)) return;
            build(ls, begin, mid);
            build(rs, mid, end);
            return;
        }
        // bin on all three axes, keep the cheapest split
        Box bins[3][BINS];
        uint32_t bcnt[3][BINS];
        float scale[3];
        for (int a = 0; a < 3; ++a) {
            const float ext = cb.hi[a] - cb.lo[a];
            scale[a] = ext > 0.0f ? (float)BINS / ext : 0.0f;
            for (int b = 0; b < BINS; ++b) {
                box_init(bins[a][b]);
                bcnt[a][b] = 0;
            }
        }
        auto bin_of = [&](uint32_t p, int a) {
            int b = (int)((cen[3 * (size_t)p + a] - cb.lo[a]) * scale[a]);
            return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
        };
        for (uint32_t i = begin; i < end; ++i) {
            const uint32_t p = order[i];
            for (int a = 0; a < 3; ++a) {
                if (scale[a] == 0.0f) continue;
                const int b = bin_of(p, a);
                bcnt[a][b]++;
                box_grow(bins[a][b], boxes[p]);
            }
        }
        float best = INFINITY;
        int best_axis = -1, best_bin = -1;
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
                if (cost < best) {
                    best = cost;
                    best_axis = a;
                    best_bin = b;
                }
            }
        }
        uint32_t mid = begin + count / 2;
        if (best_axis >= 0) {
            uint32_t i = begin, j = end;
            while (i < j) {
                const uint32_t p = order[i];
                if (bin_of(p, best_axis) <= best_bin) {
                    ++i;
                } else {
                    --j;
                    order[i] = order[j];
                    order[j] = p;
                }
            }
            if (i > begin && i < end) mid = i;
        }
        const uint32_t nl = mid - begin;
        const uint32_t ls = slot + 1, rs = slot + 2 * nl;
        n.left = ls;
        n.right = rs;
        n.count = 0;
        if (cancelled.load(std::memory_order_relaxed)) return;
        bool spawned = false;
        std::thread worker;
        if (nl >= PARALLEL_MIN && (end - mid) >= PARALLEL_MIN) {
            int cur = live.load();
            while (cur < max_threads && !live.compare_exchange_weak(cur, cur + 1)) {
            }
            if (cur < max_threads) {
                spawned = true;
                worker = std::thread([this, ls, begin, mid] {
                    build(ls, begin, mid);
                    live.fetch_sub(1);
                });
            }
        }
        if (!spawned) build(ls, begin, mid);
        build(rs, mid, end);
        if (spawned) worker.join();
    }
};

} // namespace

bool build(const std::vector<Box> &boxes, WideBvh &out, const std::function<bool(double)> &progress, unsigned threads) {
    out = WideBvh();
    const size_t n = boxes.size();
    if (progress && !progress(0.0)) return false;
    if (n == 0) {
        if (progress) progress(1.0);
        return true;
    }
    Builder b(boxes);
    b.cen.resize(3 * n);
    b.order.resize(n);
    b.nodes.resize(2 * n);
    for (size_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) b.cen[3 * i + a] = 0.5f * boxes[i].lo[a] + 0.5f * boxes[i].hi[a];
        b.order[i] = (uint32_t)i;
    }
    unsigned hw = threads ? threads : std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (hw > 16) hw = 16;
    b.max_threads = (int)hw;
    if (const char *v = getenv("RTMI_SWEEP_MAX")) b.sweep_max = (uint32_t)atoi(v);

    // run the build on a worker so that this thread can report progress
    std::atomic<bool> finished{false};
    std::thread root([&] {
        b.build(0, 0, (uint32_t)n);
        finished.store(true);
    });
    bool ok = true;
    while (!finished.load()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(n > 100000 ? 20 : 1));
        if (progress && ok && !progress(0.8 * (double)b.done.load() / (double)n)) {
            ok = false;
            b.cancelled.store(true);
        }
    }
    root.join();
    if (!ok) return false;

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
                    refs[k] = rtd::RT_LEAF | ((c.count - 1u) << 28) | c.first;
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
    out.prim_order = std::move(b.order);

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
    const uint32_t n_nodes = out.n_nodes, n_slots = (uint32_t)n;
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
