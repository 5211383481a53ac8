// scene_build.cpp -- see scene_build.h.
#include "scene_build.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

namespace rts {

namespace {

inline float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
inline float min3(float a, float b, float c) { return fminf(fminf(a, b), c); }

void prim_box(const HostPrim &p, rtb::Box &b) {
    if (p.is_sphere) {
        const float r = p.b[0];
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = p.a[k] - r;
            b.hi[k] = p.a[k] + r;
        }
    } else {
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = min3(p.a[k], p.b[k], p.c[k]);
            b.hi[k] = max3(p.a[k], p.b[k], p.c[k]);
        }
    }
}

// Culling boxes are padded so that a hit the f32 primitive test accepts is never culled by the slab test, whose
// distances are fused multiply-adds of magnitudes up to |origin| / |direction component|: a term relative to the box's
// own coordinates plus one relative to the scene's (covering ray origins within ~16 scene extents).
void pad_box(rtb::Box &b, float scene_scale) {
    float m = 0.0f;
    for (int k = 0; k < 3; ++k) m = fmaxf(m, fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k])));
    const float e = 1e-5f * m + 1e-6f * scene_scale + 1e-30f;
    for (int k = 0; k < 3; ++k) {
        b.lo[k] -= e;
        b.hi[k] += e;
    }
}

// The part of triangle (a, b, c) inside `region` (Sutherland-Hodgman against the six planes, in double): its bounds are
// merged into lo / hi; false when nothing of the triangle is inside.
bool clip_triangle(const float *a, const float *b, const float *c, const rtb::Box &region, double lo[3], double hi[3]) {
    double poly[2][12][3];
    int n = 3, cur = 0;
    for (int k = 0; k < 3; ++k) {
        poly[0][0][k] = a[k];
        poly[0][1][k] = b[k];
        poly[0][2][k] = c[k];
    }
    for (int axis = 0; axis < 3 && n > 0; ++axis)
        for (int side = 0; side < 2 && n > 0; ++side) {
            const double plane = side == 0 ? (double)region.lo[axis] : (double)region.hi[axis];
            const double sign = side == 0 ? 1.0 : -1.0;
            double(*in)[3] = poly[cur], (*out)[3] = poly[cur ^ 1];
            int m = 0;
            for (int i = 0; i < n; ++i) {
                const double *p = in[i], *q = in[(i + 1) % n];
                const double dp = sign * (p[axis] - plane), dq = sign * (q[axis] - plane);
                if (dp >= 0.0) {
                    for (int k = 0; k < 3; ++k) out[m][k] = p[k];
                    ++m;
                }
                if ((dp > 0.0 && dq < 0.0) || (dp < 0.0 && dq > 0.0)) {
                    const double t = dp / (dp - dq);
                    for (int k = 0; k < 3; ++k) out[m][k] = p[k] + t * (q[k] - p[k]);
                    out[m][axis] = plane;
                    ++m;
                }
            }
            n = m;
            cur ^= 1;
        }
    if (n == 0) return false;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            lo[k] = fmin(lo[k], poly[cur][i][k]);
            hi[k] = fmax(hi[k], poly[cur][i][k]);
        }
    return true;
}

} // namespace

void apply_transform(const float translation[3], const float scale[3], const float rot[4], float p[3]) {
    const float bx = rot[0], by = rot[1], bz = rot[2], w = rot[3];
    const float b2 = (bx * bx + by * by) + bz * bz;
    const float k = w * w - b2;
    const float d2 = ((p[0] * bx + p[1] * by) + p[2] * bz) * 2.0f;
    const float w2 = w * 2.0f;
    const float cx = by * p[2] - p[1] * bz;
    const float cy = bz * p[0] - p[2] * bx;
    const float cz = bx * p[1] - p[0] * by;
    const float rx = (p[0] * k + bx * d2) + cx * w2;
    const float ry = (p[1] * k + by * d2) + cy * w2;
    const float rz = (p[2] * k + bz * d2) + cz * w2;
    p[0] = scale[0] * rx + translation[0];
    p[1] = scale[1] * ry + translation[1];
    p[2] = scale[2] * rz + translation[2];
}

namespace {

// ---- triangle pairing ----------------------------------------------------------------------------------------

constexpr uint32_t NONE = 0xFFFFFFFFu;

struct EdgeRef {
    uint64_t key;
    uint32_t tri;
    uint32_t edge;
};

inline const float *corner(const HostPrim &p, int k) { return k == 0 ? p.a : (k == 1 ? p.b : p.c); }
inline bool same_point(const float *x, const float *y) { return memcmp(x, y, 12) == 0; }

inline uint64_t mix64(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xFF51AFD7ED558CCDull;
    return h ^ (h >> 33);
}

// key of the undirected edge between corners k and k + 1 of triangle p: geometry and both end points, bit for bit
uint64_t edge_key(const HostPrim &p, int k) {
    const float *x = corner(p, k), *y = corner(p, (k + 1) % 3);
    if (memcmp(x, y, 12) > 0) std::swap(x, y);
    uint32_t w[6];
    memcpy(w, x, 12);
    memcpy(w + 3, y, 12);
    uint64_t h = p.geom;
    for (int i = 0; i < 6; i += 2) h = mix64(h, ((uint64_t)w[i] << 32) | w[i + 1]);
    return h;
}

bool same_edge(const HostPrim &p, int pe, const HostPrim &q, int qe) {
    if (p.geom != q.geom) return false;
    const float *p0 = corner(p, pe), *p1 = corner(p, (pe + 1) % 3);
    const float *q0 = corner(q, qe), *q1 = corner(q, (qe + 1) % 3);
    return (same_point(p0, q0) && same_point(p1, q1)) || (same_point(p0, q1) && same_point(p1, q0));
}

// Sorts edge references by key: bucket by the top byte, then sort the buckets on worker threads.
void sort_edges(std::vector<EdgeRef> &v, unsigned threads) {
    const size_t n = v.size();
    if (n < (1u << 16) || threads < 2) {
        std::sort(v.begin(), v.end(), [](const EdgeRef &x, const EdgeRef &y) { return x.key < y.key; });
        return;
    }
    std::vector<size_t> start(257, 0);
    for (const EdgeRef &e : v) ++start[(e.key >> 56) + 1];
    for (int b = 0; b < 256; ++b) start[b + 1] += start[b];
    std::vector<EdgeRef> tmp(n);
    {
        std::vector<size_t> at(start.begin(), start.end() - 1);
        for (const EdgeRef &e : v) tmp[at[e.key >> 56]++] = e;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            for (int b; (b = next.fetch_add(1)) < 256;)
                std::sort(tmp.begin() + start[b], tmp.begin() + start[b + 1],
                          [](const EdgeRef &x, const EdgeRef &y) { return x.key < y.key; });
        });
    for (auto &th : pool) th.join();
    v.swap(tmp);
}

// mate[i] = the triangle that shares a record with triangle i, or NONE.  Two triangles of one geometry that share an
// edge (same end points, bit for bit) have at most four distinct corners between them; among the free neighbours of a
// triangle the one with the smallest common box is taken.
void pair_triangles(const std::vector<HostPrim> &prims, const std::vector<rtb::Box> &boxes, std::vector<uint32_t> &mate) {
    const size_t n = prims.size();
    mate.assign(n, NONE);
    std::vector<EdgeRef> edges;
    edges.reserve(3 * n);
    for (size_t i = 0; i < n; ++i) {
        if (prims[i].is_sphere) continue;
        for (int k = 0; k < 3; ++k) edges.push_back(EdgeRef{edge_key(prims[i], k), (uint32_t)i, (uint32_t)k});
    }
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (hw > 16) hw = 16;
    sort_edges(edges, hw);
    std::vector<uint32_t> adj(3 * n, NONE);
    for (size_t k = 0; k + 1 < edges.size();) {
        const EdgeRef &x = edges[k], &y = edges[k + 1];
        if (x.key == y.key && x.tri != y.tri && same_edge(prims[x.tri], (int)x.edge, prims[y.tri], (int)y.edge)) {
            adj[3 * (size_t)x.tri + x.edge] = y.tri;
            adj[3 * (size_t)y.tri + y.edge] = x.tri;
            k += 2;
        } else {
            ++k;
        }
    }
    for (size_t i = 0; i < n; ++i) {
        if (mate[i] != NONE || prims[i].is_sphere) continue;
        uint32_t best = NONE;
        float best_area = INFINITY;
        for (int k = 0; k < 3; ++k) {
            const uint32_t j = adj[3 * i + k];
            if (j == NONE || mate[j] != NONE) continue;
            float area = 0.0f, ext[3];
            for (int a = 0; a < 3; ++a)
                ext[a] = fmaxf(boxes[i].hi[a], boxes[j].hi[a]) - fminf(boxes[i].lo[a], boxes[j].lo[a]);
            area = ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0];
            if (!(area >= best_area)) {
                best_area = area;
                best = j;
            }
        }
        if (best != NONE) {
            mate[i] = best;
            mate[best] = (uint32_t)i;
        }
    }
}

} // namespace

bool build_scene(const std::vector<HostPrim> &prims, rtb::WideBvh &bvh, size_t *n_records, float lo[3], float hi[3],
                 const std::function<bool(double)> &progress) {
    const size_t n = prims.size();
    std::vector<rtb::Box> boxes(n);
    for (int k = 0; k < 3; ++k) {
        lo[k] = INFINITY;
        hi[k] = -INFINITY;
    }
    for (size_t i = 0; i < n; ++i) {
        prim_box(prims[i], boxes[i]);
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], boxes[i].lo[k]);
            hi[k] = fmaxf(hi[k], boxes[i].hi[k]);
        }
    }
    float scene_scale = 0.0f;
    for (int k = 0; k < 3; ++k)
        if (n) scene_scale = fmaxf(scene_scale, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
    for (size_t i = 0; i < n; ++i) pad_box(boxes[i], scene_scale);
    if (progress && !progress(0.0)) return false;

    // records: a triangle with its mate, a lone triangle, or a sphere; in insertion order of their first primitive
    std::vector<uint32_t> mate;
    pair_triangles(prims, boxes, mate);
    std::vector<uint32_t> rec_first;
    rec_first.reserve(n / 2 + 16);
    for (size_t i = 0; i < n; ++i)
        if (mate[i] == NONE || mate[i] > i) rec_first.push_back((uint32_t)i);
    const size_t nr = rec_first.size();
    std::vector<rtb::Box> rec_boxes(nr);
    for (size_t r = 0; r < nr; ++r) {
        const uint32_t i = rec_first[r], j = mate[i];
        rec_boxes[r] = boxes[i];
        if (j != NONE)
            for (int k = 0; k < 3; ++k) {
                rec_boxes[r].lo[k] = fminf(rec_boxes[r].lo[k], boxes[j].lo[k]);
                rec_boxes[r].hi[k] = fmaxf(rec_boxes[r].hi[k], boxes[j].hi[k]);
            }
    }
    // first global primitive index of every geometry (primitives of a geometry are consecutive)
    std::vector<uint32_t> geom_first;
    for (size_t i = 0; i < n; ++i) {
        if (prims[i].geom >= geom_first.size()) geom_first.resize(prims[i].geom + 1, NONE);
        if (geom_first[prims[i].geom] == NONE) geom_first[prims[i].geom] = (uint32_t)i;
    }
    // What a spatial split of the builder asks for: the padded bounds of the part of record r inside a region.  A
    // triangle is clipped exactly, a sphere by its box; the result is rounded outwards to f32 and padded like the
    // primitive boxes, so a part keeps the margin the slab test relies on.
    const rtb::ClipFn clip = [&](uint32_t r, const rtb::Box &region, rtb::Box &out) {
        const uint32_t i = rec_first[r], j = mate[i];
        double dlo[3] = {INFINITY, INFINITY, INFINITY}, dhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool any = false;
        if (prims[i].is_sphere) {
            rtb::Box sb;
            prim_box(prims[i], sb);
            any = true;
            for (int k = 0; k < 3; ++k) {
                dlo[k] = fmax((double)sb.lo[k], (double)region.lo[k]);
                dhi[k] = fmin((double)sb.hi[k], (double)region.hi[k]);
                if (!(dlo[k] <= dhi[k])) any = false;
            }
        } else {
            any = clip_triangle(prims[i].a, prims[i].b, prims[i].c, region, dlo, dhi);
            if (j != NONE) any = clip_triangle(prims[j].a, prims[j].b, prims[j].c, region, dlo, dhi) || any;
        }
        if (!any) return false;
        for (int k = 0; k < 3; ++k) {
            out.lo[k] = (float)dlo[k];
            out.hi[k] = (float)dhi[k];
            if ((double)out.lo[k] > dlo[k]) out.lo[k] = nextafterf(out.lo[k], -INFINITY);
            if ((double)out.hi[k] < dhi[k]) out.hi[k] = nextafterf(out.hi[k], INFINITY);
        }
        pad_box(out, scene_scale);
        return true;
    };
    if (!rtb::build(rec_boxes, bvh, progress, 0, clip)) return false;
    if (n_records) *n_records = bvh.prim_order.size();
    for (size_t slot = 0; slot < bvh.prim_order.size(); ++slot) {
        const uint32_t i = rec_first[bvh.prim_order[slot]], j = mate[i];
        const HostPrim &p = prims[i];
        float w[16];
        for (int k = 0; k < 16; ++k) w[k] = 0.0f;
        uint32_t ids[4] = {i, NONE, p.geom & rtd::RT_GEOM_MASK, geom_first[p.geom]};
        if (p.is_sphere) {
            memcpy(&w[0], p.a, 12);
            w[3] = p.b[0];
            ids[2] |= rtd::RT_SPHERE;
        } else {
            memcpy(&w[0], p.a, 12);
            memcpy(&w[3], p.b, 12);
            memcpy(&w[6], p.c, 12);
            if (j != NONE) {
                const HostPrim &q = prims[j];
                uint32_t code = 0;
                bool have_d = false, ok = true;
                for (int k = 0; k < 3; ++k) {
                    const float *x = corner(q, k);
                    uint32_t sel;
                    if (same_point(x, p.a))
                        sel = 0;
                    else if (same_point(x, p.b))
                        sel = 1;
                    else if (same_point(x, p.c))
                        sel = 2;
                    else if (!have_d || same_point(x, &w[9])) {
                        memcpy(&w[9], x, 12);
                        have_d = true;
                        sel = 3;
                    } else {
                        ok = false;
                        sel = 0;
                    }
                    code |= sel << (2 * k);
                }
                if (!ok) return false; // cannot happen: mates share an edge
                ids[1] = j;
                ids[2] |= code << 24;
            }
        }
        memcpy(&w[12], ids, 16);
        memcpy(&bvh.items[(size_t)bvh.record_item[slot] * rtd::RT_PRIM_F4], w, 64);
    }
    return true;
}

} // namespace rts
