// scene_build.cpp -- see scene_build.h.
#include "scene_build.h"

#include <math.h>
#include <string.h>

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

bool build_scene(const std::vector<HostPrim> &prims, rtb::WideBvh &bvh, std::vector<rtd::F4> &records, float lo[3],
                 float hi[3], const std::function<bool(double)> &progress) {
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
    if (!rtb::build(boxes, bvh, progress)) return false;
    records.assign(n * rtd::RT_PRIM_F4, rtd::F4{0.0f, 0.0f, 0.0f, 0.0f});
    for (size_t slot = 0; slot < n; ++slot) {
        const uint32_t gi = bvh.prim_order[slot];
        const HostPrim &p = prims[gi];
        float w[12];
        if (p.is_sphere) {
            w[0] = p.a[0];
            w[1] = p.a[1];
            w[2] = p.a[2];
            w[3] = p.b[0];
            for (int k = 4; k < 9; ++k) w[k] = 0.0f;
        } else {
            for (int k = 0; k < 3; ++k) {
                w[k] = p.a[k];
                w[3 + k] = p.b[k] - p.a[k];
                w[6 + k] = p.c[k] - p.a[k];
            }
        }
        const uint32_t ids[3] = {gi, p.geom | (p.is_sphere ? rtd::RT_SPHERE : 0u), p.prim};
        memcpy(&w[9], ids, 12);
        memcpy(&records[slot * rtd::RT_PRIM_F4], w, 48);
    }
    return true;
}

} // namespace rts
