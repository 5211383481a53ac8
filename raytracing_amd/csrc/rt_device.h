// rt_device.h -- per-ray device code of the path-tracing kernels: closest-hit traversal of the flattened 4-wide
// BVH, Moeller-Trumbore / sphere tests, counter-based RNG, camera rays, BSDF sampling and the per-bounce shading
// step.  Every function is straight-line f32 arithmetic in a fixed operation order (build with
// -ffp-contract=off), so a path's result does not depend on which lane, wave or device evaluates it.
//
// Reference behaviour reproduced here (paths relative to the reference checkout):
//   closest hit contract   crates/rt/src/aggregate/embree.rs:100-140
//   path tracer            crates/rt/src/integrators/pathtracing.rs:21-79, integrators/mod.rs:8-23
//   random walk            crates/rt/src/integrators/randomwalk.rs:24-68
//   BSDF / BxDFs           crates/rt/src/material/mod.rs:56-394
//   distributions          crates/rt/src/math/distributions.rs:86-95,120-130,176-265
//   frame                  crates/rt/src/math/transform.rs:63-94
//   reflect / refract      crates/rt/src/math/vec.rs:14-52
//   camera                 crates/rt/src/camera.rs:62-107
//   sampler / filter       crates/rt/src/sampler.rs:91-102, crates/rt/src/filter.rs:23-32
//   pixel driver           crates/rt-cli/src/executor.rs:254-270
//
// RT_DEV is `__device__ __forceinline__` in the kernels.  A host debugging build may define it as `inline` to
// single-step the same code under gdb; no shipped code path does.
#pragma once

#include <math.h>
#include <stdint.h>

#ifndef RT_DEV
#define RT_DEV __device__ __forceinline__
#endif

namespace rtd {

// ---------------------------------------------------------------------------------------------- layout

// BVH4 node: a 64-byte item of which the kernels read the first 48 bytes (three 16-byte loads instead of four: the
// traversal kernels are bound by the vector memory pipeline on incoherent rays, where one more 16-byte load per node
// step measured +20-25 %).  Words: 0-2 origin.xyz; 3 = the upper halves of step.x | step.y << 16 (steps are powers of
// two, so the upper half of the float is the float); 4-9 six words of 8-bit child planes (lo.x, lo.y, lo.z, hi.x, hi.y,
// hi.z; child k in byte k; plane = origin + q * step, rounded outwards at build time); 10 = upper half of step.z |
// leaf bits << 16 (bit k: child k is a leaf) | (children - 1) << 24; 11 = item index of child 0, child k being item
// first + k (the children of a node are neighbours in the item array, and fill slots 0 .. children - 1).  Words 12-15
// repeat the four child refs in the form below for host-side tools; no kernel reads them.
// Child ref (traversal stack entries, Trav::cur): RT_EMPTY = none; bit 31 set = leaf: bits 28..30 = records - 1,
// bits 0..27 = item of the first record; otherwise the item index of an inner node.
struct alignas(16) F4 {
    float x, y, z, w;
};
struct alignas(16) U4 {
    uint32_t x, y, z, w;
};
constexpr uint32_t RT_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t RT_LEAF = 0x80000000u;
constexpr uint32_t RT_SPHERE = 0x80000000u;
constexpr int RT_NODE_F4 = 4; // float4s per node
constexpr uint32_t RT_GEOM_MASK = 0x00FFFFFFu;
constexpr int RT_PRIM_F4 = 4; // float4s per primitive record

// 64-byte primitive record, 16 words: four points a, b, c, d (12 words), then the global primitive index (insertion
// order) of the first and of the second triangle (RT_EMPTY = none), geometry id | selectors << 24 | RT_SPHERE, and the
// global index of the geometry's first primitive.  The first triangle is (a, b, c) in its own vertex order; the second
// shares two corners with it and names its corners by three 2-bit selectors into (a, b, c, d), again in its own
// order, so that both are tested on exactly the edge vectors v1 - v0, v2 - v0 of the triangle as it was inserted.
// A sphere keeps its centre in a and its radius in b.x.

struct Material {
    uint32_t kind;
    float r, g, b;
    float ior;
    float roughness;
    uint32_t pad0, pad1;
};
constexpr uint32_t MAT_DIFFUSE = 0, MAT_DIELECTRIC = 1, MAT_THIN = 2, MAT_EMIT = 3;

struct SceneView {
    const F4 *nodes;
    const F4 *prims;
    const uint32_t *geom_material;
    const Material *materials;
    uint32_t n_nodes;
    uint32_t n_prims;
    uint32_t n_geoms;
    uint32_t n_materials;
};

// Node and record addresses are a uniform base plus a 32-bit byte offset (one shift; the hardware adds the base), which
// is why a scene is limited to RT_MAX_ITEMS items (nodes + records + the padding of the placement).
constexpr uint32_t RT_MAX_ITEMS = 1u << 26;
RT_DEV const U4 *node_at(const SceneView &sv, uint32_t node) {
    return reinterpret_cast<const U4 *>(reinterpret_cast<const char *>(sv.nodes) + (node << 6));
}
RT_DEV const F4 *record_at(const SceneView &sv, uint32_t slot) {
    return reinterpret_cast<const F4 *>(reinterpret_cast<const char *>(sv.prims) + (slot << 6));
}

// Makes the compiler treat a loaded 16-byte word as used as a whole, right here.  Without it the record fetch of
// test_prim, whose words are consumed on two sides of the sphere / triangle branch, is compiled to seven loads of
// one to four dwords scattered over both sides instead of four 16-byte ones (measured: 2.5 % of a frame).
RT_DEV void keep_whole(F4 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
#else
    (void)v;
#endif
}

// Conditions and selects.  For `c ? a : b` the compiler emits v_cndmask_b32 with the condition in vcc, and a wave that
// issues two of those in a row waits some fifteen cycles for the second (measured, profiles/experiments/
// pk_fma_probe.hip: runs of selects on vcc issue at a quarter of the rate of the same selects on a scalar register
// pair); traversal code is full of such runs (near / far planes, sorting network, best-candidate update).  So the hot
// code keeps its conditions as lane masks in scalar registers -- lanes(c), combined with & and | on the scalar unit --
// and selects through them.  On the host a lane mask is 0 or 1.
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long LaneMask;
RT_DEV LaneMask lanes(bool c) { return __builtin_amdgcn_ballot_w64(c); }
RT_DEV float select(LaneMask m, float a, float b) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
RT_DEV uint32_t select(LaneMask m, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
#else
typedef unsigned LaneMask; // 0 or 1
RT_DEV LaneMask lanes(bool c) { return c ? 1u : 0u; }
RT_DEV float select(LaneMask m, float a, float b) { return m ? a : b; }
RT_DEV uint32_t select(LaneMask m, uint32_t a, uint32_t b) { return m ? a : b; }
#endif

struct CameraView {
    uint32_t width, height;
    float half_width, half_height, focal_length, aperture;
    float lens[3];
    float rot[4];
};

// ---------------------------------------------------------------------------------------------- vectors

struct V3 {
    float x, y, z;
};
RT_DEV V3 mk(float x, float y, float z) {
    V3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
RT_DEV V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 scale(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
RT_DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
RT_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) { return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
RT_DEV V3 normalize(V3 a) {
    float rcp = 1.0f / sqrtf(dot(a, a));
    return mk(a.x * rcp, a.y * rcp, a.z * rcp);
}
RT_DEV V3 normalize_or_zero(V3 a) {
    float rcp = 1.0f / sqrtf(dot(a, a));
    if (isfinite(rcp) && rcp > 0.0f) return mk(a.x * rcp, a.y * rcp, a.z * rcp);
    return mk(0.0f, 0.0f, 0.0f);
}
RT_DEV float max_element(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }

constexpr float RT_PI = 3.14159265358979323846f;
constexpr float RT_FRAC_1_PI = 0.318309886183790671538f;
constexpr float RT_TNEAR = 0.00001f; // pathtracing.rs:29

// ---------------------------------------------------------------------------------------------- RNG

// Philox4x32-10 keyed by the 64-bit seed; counter = (x, y, sample, stream).  stream 0 = camera sample,
// stream 1 + d = bounce d.  Floats are the top 23 bits as k * 2^-23 (rand 0.8.5's Uniform<f32>(0, 1)).
RT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

RT_DEV void uniform4(uint64_t seed, uint32_t x, uint32_t y, uint32_t sample, uint32_t stream, float u[4]) {
    uint32_t r[4];
    philox4x32_10(x, y, sample, stream, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (float)(r[i] >> 9) * 1.1920928955078125e-07f;
}

// sin(2 pi u), cos(2 pi u), u in [0,1): exact removal of the nearest quarter turn, then Taylor polynomials in
// x = 2 pi r (|x| <= pi/4) by Horner in plain multiply/add.
RT_DEV void sincos_2pi(float u, float &s, float &c) {
    float qf = floorf(u * 4.0f + 0.5f);
    float r = u - qf * 0.25f;
    float x = 6.28318530717958647692f * r;
    float x2 = x * x;
    float sp = x * (1.0f + x2 * (-1.66666666666666666667e-01f +
                                 x2 * (8.33333333333333333333e-03f +
                                       x2 * (-1.98412698412698412698e-04f + x2 * 2.75573192239858906526e-06f))));
    float cp = 1.0f + x2 * (-0.5f + x2 * (4.16666666666666666667e-02f +
                                          x2 * (-1.38888888888888888889e-03f +
                                                x2 * (2.48015873015873015873e-05f +
                                                      x2 * -2.75573192239858906526e-07f))));
    int q = (int)qf & 3;
    if (q == 0) {
        s = sp;
        c = cp;
    } else if (q == 1) {
        s = cp;
        c = -sp;
    } else if (q == 2) {
        s = -sp;
        c = -cp;
    } else {
        s = -cp;
        c = sp;
    }
}

// ---------------------------------------------------------------------------------------------- frame

struct Frame {
    V3 c0, c1, c2;
};
// transform.rs:63-86
RT_DEV Frame frame_new(V3 n) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    Frame f;
    f.c0 = mk(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    f.c1 = mk(b, sign + n.y * n.y * a, -n.y);
    f.c2 = n;
    return f;
}
RT_DEV V3 to_local(const Frame &f, V3 g) { return mk(dot(f.c0, g), dot(f.c1, g), dot(f.c2, g)); }
RT_DEV V3 from_local(const Frame &f, V3 l) {
    return mk((f.c0.x * l.x + f.c1.x * l.y) + f.c2.x * l.z, (f.c0.y * l.x + f.c1.y * l.y) + f.c2.y * l.z,
              (f.c0.z * l.x + f.c1.z * l.y) + f.c2.z * l.z);
}

// ---------------------------------------------------------------------------------------------- sampling

// distributions.rs:86-95
RT_DEV void ball2(float u0, float u1, float &px, float &py) {
    float s, c;
    sincos_2pi(u0, s, c);
    float r = sqrtf(u1);
    px = r * c;
    py = r * s;
}
// distributions.rs:176-188
RT_DEV V3 cosine_hemisphere(float u0, float u1) {
    float px, py;
    ball2(u0, u1, px, py);
    float z = sqrtf(1.0f - px * px - py * py);
    return mk(px, py, z);
}
// distributions.rs:120-130 (theta = pi * u1 evaluated as 2 pi (u1 / 2))
RT_DEV V3 uniform_unit_sphere3(float u0, float u1) {
    float sphi, cphi, stheta, ctheta;
    sincos_2pi(u0, sphi, cphi);
    sincos_2pi(u1 * 0.5f, stheta, ctheta);
    return mk(cphi * stheta, sphi * stheta, ctheta);
}
// float.rs:26-28
RT_DEV float lerp_ext(float self, float from, float to) { return self * to + (1.0f - self) * from; }
// vec.rs:16-18
RT_DEV bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.0f; }
// vec.rs:27-29
RT_DEV V3 reflect(V3 w, V3 n) { return add(neg(w), scale(2.0f * dot(w, n), n)); }
// vec.rs:33-52
RT_DEV bool refract(V3 w, V3 normal, float ior, V3 &out, float &ior_out) {
    float cosi = dot(w, normal);
    if (!(cosi >= 0.0f)) {
        normal = neg(normal);
        cosi = -cosi;
        ior = 1.0f / ior;
    }
    float cost2 = 1.0f - fmaxf(0.0f, 1.0f - cosi * cosi) / ior / ior;
    if (cost2 <= 0.0f) return false;
    float cost = sqrtf(cost2);
    V3 a = mk(-w.x / ior, -w.y / ior, -w.z / ior);
    out = add(a, scale(cosi / ior - cost, normal));
    ior_out = ior;
    return true;
}
// material/mod.rs:139-154
RT_DEV float fresnel_dielectric(float cosi, float ior) {
    if (!(cosi >= 0.0f)) {
        cosi = -cosi;
        ior = 1.0f / ior;
    }
    float cost2 = 1.0f - (1.0f - cosi * cosi) / ior / ior;
    if (cost2 < 0.0f) return 1.0f;
    float cost = sqrtf(cost2);
    float r_parl = (ior * cosi - cost) / (ior * cosi + cost);
    float r_perp = (cosi - ior * cost) / (cosi + ior * cost);
    return 0.5f * (r_parl * r_parl + r_perp * r_perp);
}

// distributions.rs:203-265; the normal distribution and the masking term are the reference's own variants.
RT_DEV bool tr_is_smooth(float alpha) { return alpha < 1e-8f; }
RT_DEV float tr_d(float alpha, V3 wm) {
    float cos2 = wm.z * wm.z;
    float sin2 = 1.0f - cos2;
    float tan2 = sin2 / cos2;
    float at = alpha + tan2;
    return 1.0f / (RT_PI * (cos2 * cos2) * (at * at));
}
RT_DEV float tr_lambda(float alpha, V3 w) {
    float cos2 = w.z * w.z;
    float sin2 = 1.0f - cos2;
    float tan2 = sin2 / cos2;
    return sqrtf(1.0f + (alpha * alpha) * tan2) / 2.0f;
}
RT_DEV float tr_g(float alpha, V3 w0, V3 w1) { return 1.0f / (1.0f + tr_lambda(alpha, w0) + tr_lambda(alpha, w1)); }
RT_DEV float tr_g1(float alpha, V3 w) { return 1.0f / (1.0f + tr_lambda(alpha, w)); }
RT_DEV float tr_dw(float alpha, V3 w, V3 wm) {
    return tr_g1(alpha, w) / fabsf(w.z) * tr_d(alpha, wm) * fabsf(dot(w, wm));
}
RT_DEV V3 tr_sample_wm(float alpha, V3 w, float u0, float u1) {
    w = normalize(mk(alpha * w.x, alpha * w.y, w.z));
    if (!(w.z >= 0.0f)) w = neg(w);
    V3 t1 = (w.z < 0.9999f) ? normalize(cross(mk(0.0f, 0.0f, 1.0f), w)) : mk(1.0f, 0.0f, 0.0f);
    V3 t2 = cross(w, t1);
    float px, py;
    ball2(u0, u1, px, py);
    float h = sqrtf(1.0f - px * px);
    py = lerp_ext((1.0f + w.z) / 2.0f, h, py);
    float pz = sqrtf(fmaxf(0.0f, 1.0f - (px * px + py * py)));
    V3 nh = add(add(scale(px, t1), scale(py, t2)), scale(pz, w));
    return normalize(mk(alpha * nh.x, alpha * nh.y, fmaxf(1e-6f, nh.z)));
}

// ---------------------------------------------------------------------------------------------- BxDFs

struct BxdfSample {
    V3 wi;
    V3 f;
    float pdf;
};
RT_DEV V3 white(float s) { return mk(s * 1.0f, s * 1.0f, s * 1.0f); }

// material/mod.rs:119-130
RT_DEV bool diffuse_sample_f(const Material &m, V3 wo, float u0, float u1, BxdfSample &o) {
    V3 wi = cosine_hemisphere(u0, u1);
    wi.z = copysignf(wi.z, wo.z);
    o.wi = wi;
    o.f = mk(RT_FRAC_1_PI * m.r, RT_FRAC_1_PI * m.g, RT_FRAC_1_PI * m.b);
    o.pdf = fabsf(wi.z) * RT_FRAC_1_PI;
    return true;
}

// material/mod.rs:244-312
RT_DEV bool dielectric_sample_f(const Material &m, V3 wo, float u0, float u1, float u2, BxdfSample &o) {
    float alpha = m.roughness;
    if (m.ior == 1.0f || tr_is_smooth(alpha)) {
        float r = fresnel_dielectric(wo.z, m.ior);
        float t = 1.0f - r;
        if (u2 <= r / (r + t)) {
            V3 wi = mk(-wo.x, -wo.y, wo.z);
            o.wi = wi;
            o.f = white(r / fabsf(wi.z));
            o.pdf = r / (r + t);
            return true;
        }
        V3 wi;
        float ior2;
        if (!refract(wo, mk(0.0f, 0.0f, 1.0f), m.ior, wi, ior2)) return false;
        o.wi = wi;
        o.f = white(t / fabsf(wi.z));
        o.pdf = t / (r + t);
        return true;
    }
    V3 wm = tr_sample_wm(alpha, wo, u0, u1);
    float r = fresnel_dielectric(dot(wo, wm), m.ior);
    float t = 1.0f - r;
    // What the reflected and the refracted branch both evaluate is evaluated once, ahead of the branch: lanes of a wave
    // rarely agree on the branch, so the wave runs both, and these are the expensive terms (divisions, square roots).
    // Each value is the expression the branches use, operand for operand, so no result changes.
    const float pr = r / (r + t);
    const float d_wm = tr_d(alpha, wm);
    const float lam_o = tr_lambda(alpha, wo);
    const float dw = (1.0f / (1.0f + lam_o)) / fabsf(wo.z) * d_wm * fabsf(dot(wo, wm)); // tr_dw(alpha, wo, wm)
    if (u2 < pr) {
        V3 wi = reflect(wo, wm);
        if (!same_hemisphere(wo, wi)) return false;
        const float g = 1.0f / (1.0f + lam_o + tr_lambda(alpha, wi)); // tr_g(alpha, wo, wi)
        float f = d_wm * g * r / (4.0f * wi.z * wo.z);
        float pdf = dw / (4.0f * fabsf(dot(wo, wm))) * r / (r + t);
        o.wi = wi;
        o.f = white(f);
        o.pdf = pdf;
        return true;
    }
    V3 wi;
    float ior2;
    if (!refract(wo, wm, m.ior, wi, ior2)) return false;
    if (same_hemisphere(wi, wo) || wi.z == 0.0f) return false;
    float s = dot(wi, wm) + dot(wo, wm) / ior2;
    float denom = s * s;
    float dwm_dwi = fabsf(dot(wi, wm)) / denom;
    const float g = 1.0f / (1.0f + lam_o + tr_lambda(alpha, wi)); // tr_g(alpha, wo, wi)
    float f = t * d_wm * g * fabsf(dot(wo, wm) * dot(wi, wm) / (wi.z * wo.z * denom));
    float pdf = dw * dwm_dwi * t / (r + t);
    o.wi = wi;
    o.f = white(f);
    o.pdf = pdf;
    return true;
}

// material/mod.rs:339-366
RT_DEV bool thin_dielectric_sample_f(const Material &m, V3 wo, float u2, BxdfSample &o) {
    float r = fresnel_dielectric(wo.z, m.ior);
    float t = 1.0f - r;
    if (r >= 1.0f) {
        r = 1.0f;
        t = 0.0f;
    } else {
        r = r + t * t * r / (1.0f - r * r);
        t = 1.0f - r;
    }
    if (u2 <= r / (r + t)) {
        V3 wi = mk(-wo.x, -wo.y, wo.z);
        o.wi = wi;
        o.f = white(r / fabsf(wi.z));
        o.pdf = r / (r + t);
    } else {
        V3 wi = neg(wo);
        o.wi = wi;
        o.f = white(t / fabsf(wi.z));
        o.pdf = t / (r + t);
    }
    return true;
}

// material/mod.rs:167-207 (evaluated only by the random-walk integrator)
RT_DEV V3 dielectric_f(const Material &m, V3 wo, V3 wi) {
    float alpha = m.roughness;
    V3 black = mk(0.0f, 0.0f, 0.0f);
    if (m.ior == 1.0f || tr_is_smooth(alpha)) return black;
    float cosi = wi.z, coso = wo.z;
    bool is_reflect = cosi * coso > 0.0f;
    float ior = (is_reflect || coso > 0.0f) ? m.ior : 1.0f / m.ior;
    V3 wm = add(scale(ior, wi), wo);
    if (cosi == 0.0f || coso == 0.0f || dot(wm, wm) == 0.0f) return black;
    float sg = isnan(wm.z) ? wm.z : copysignf(1.0f, wm.z);
    wm = scale(sg, normalize(wm));
    if (dot(wi, wm) * cosi < 0.0f || dot(wo, wm) * coso < 0.0f) return black;
    float r = fresnel_dielectric(dot(wo, wm), ior);
    float t = 1.0f - r;
    if (is_reflect) return white(tr_d(alpha, wm) * tr_g(alpha, wo, wi) * r / fabsf(4.0f * cosi * coso));
    float s = dot(wi, wm) + dot(wo, wm) / ior;
    float denom = s * s * cosi * coso;
    return white(tr_d(alpha, wm) * t * tr_g(alpha, wo, wi) * fabsf(dot(wi, wm) * dot(wo, wm) / denom));
}

// BSDF::sample_f, material/mod.rs:76-86
RT_DEV bool bsdf_sample_f(const Material &m, V3 n, V3 wo, float u0, float u1, float u2, BxdfSample &o) {
    Frame fr = frame_new(n);
    V3 wo_local = to_local(fr, wo);
    if (wo_local.z == 0.0f) return false;
    bool ok;
    if (m.kind == MAT_DIFFUSE)
        ok = diffuse_sample_f(m, wo_local, u0, u1, o);
    else if (m.kind == MAT_DIELECTRIC)
        ok = dielectric_sample_f(m, wo_local, u0, u1, u2, o);
    else if (m.kind == MAT_THIN)
        ok = thin_dielectric_sample_f(m, wo_local, u2, o);
    else
        ok = false; // EmitBxDF::sample_f -> None, material/mod.rs:387-389
    if (!ok) return false;
    o.wi = from_local(fr, o.wi);
    return true;
}

// BSDF::f, material/mod.rs:88-91
RT_DEV V3 bsdf_f(const Material &m, V3 n, V3 wo, V3 wi) {
    Frame fr = frame_new(n);
    V3 wol = to_local(fr, wo);
    V3 wil = to_local(fr, wi);
    if (m.kind == MAT_DIFFUSE) {
        if (!same_hemisphere(wol, wil)) return mk(0.0f, 0.0f, 0.0f);
        return mk(RT_FRAC_1_PI * m.r, RT_FRAC_1_PI * m.g, RT_FRAC_1_PI * m.b);
    }
    if (m.kind == MAT_DIELECTRIC) return dielectric_f(m, wol, wil);
    return mk(0.0f, 0.0f, 0.0f);
}

RT_DEV V3 material_le(const Material &m) {
    if (m.kind == MAT_EMIT) return mk(m.r, m.g, m.b);
    return mk(0.0f, 0.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------- camera

// glam Quat * Vec3
RT_DEV V3 quat_mul_vec3(const float q[4], V3 v) {
    float w = q[3];
    V3 b = mk(q[0], q[1], q[2]);
    float b2 = dot(b, b);
    return add(add(scale(w * w - b2, v), scale(dot(v, b) * 2.0f, b)), scale(w * 2.0f, cross(b, v)));
}

// Pixel sampler and filter selection (low byte of the render flags).  The executor only ever builds a
// StratifiedSampler and a BoxFilter of radius 0.7 (executor.rs:217-223); the other kinds of sampler.rs and filter.rs
// are offered for completeness.
constexpr uint32_t RT_SAMPLER_STRATIFIED = 0u, RT_SAMPLER_UNIFORM = 1u, RT_SAMPLER_PIXEL_CENTRE = 2u; // bits 0-1
constexpr uint32_t RT_FILTER_BOX = 0u, RT_FILTER_TRIANGLE = 1u, RT_FILTER_NONE = 2u;                  // bits 4-5

// filter.rs:40-47
RT_DEV float sample_tent(float c) { return c <= 0.5f ? sqrtf(2.0f * c) : 1.0f - sqrtf(2.0f - 2.0f * c); }

// filter.rs:11-55 with radius 0.7 on both axes
RT_DEV float filter_offset(uint32_t kind, float p) {
    if (kind == RT_FILTER_TRIANGLE) return 0.7f * (2.0f * sample_tent(p) - 1.0f);
    if (kind == RT_FILTER_NONE) return p;
    return lerp_ext(p, -0.7f, 0.7f);
}

// executor.rs:254-267 + sampler.rs:18-112 + filter.rs:11-55 + camera.rs:62-85.  strata = floor(sqrt(spp)).
RT_DEV void camera_ray(const CameraView &cam, uint32_t strata, uint32_t flags, uint64_t seed, uint32_t x, uint32_t y,
                       uint32_t sample, V3 &org, V3 &dir) {
    const uint32_t sampler = flags & 3u, filter = (flags >> 4) & 3u;
    uint32_t n = strata;
    uint32_t index = sample % (n * n);
    float cx = (float)(index % n), cy = (float)(index / n);
    float r[4];
    uniform4(seed, x, y, sample, 0u, r);
    // Sampler::sample_2d, twice: pixel position, then lens position
    float px, py, dx, dy;
    if (sampler == RT_SAMPLER_UNIFORM) {
        px = r[0], py = r[1], dx = r[2], dy = r[3];
    } else if (sampler == RT_SAMPLER_PIXEL_CENTRE) {
        px = py = dx = dy = 0.5f;
    } else {
        px = (cx + r[0]) / (float)n, py = (cy + r[1]) / (float)n;
        dx = (cx + r[2]) / (float)n, dy = (cy + r[3]) / (float)n;
    }
    float offx = filter_offset(filter, px), offy = filter_offset(filter, py);
    float cxp = ((float)x + 0.5f) + offx, cyp = ((float)y + 0.5f) + offy;
    float vx = 2.0f * (cxp / (float)cam.width) - 1.0f;
    float vy = 2.0f * (cyp / (float)cam.height) - 1.0f;
    V3 lens = mk(cam.lens[0], cam.lens[1], cam.lens[2]);
    V3 sensor = mk(lens.x + vx * cam.half_width, lens.y + vy * cam.half_height, lens.z + cam.focal_length);
    float ha = cam.aperture / 2.0f;
    V3 dst = mk(lens.x + ha * dx, lens.y + ha * dy, lens.z + ha * 0.0f);
    org = lens;
    dir = normalize(quat_mul_vec3(cam.rot, sub(dst, sensor)));
}

// ---------------------------------------------------------------------------------------------- closest hit

struct Closest {
    float t;
    uint32_t gidx; // global primitive index of the best candidate, RT_EMPTY = none
    uint32_t slot; // 2 * its record slot + its place in the record
    float u, v;
};

// Candidate order: smaller t wins, equal t goes to the smaller global primitive index, so the result is the same
// for every traversal order.
RT_DEV void take_candidate(float t, uint32_t gidx, uint32_t slot, float u, float v, LaneMask ok, Closest &best) {
    const LaneMask take = ok & (lanes(t < best.t) | (lanes(t == best.t) & lanes(gidx < best.gidx)));
    best.t = select(take, t, best.t);
    best.gidx = select(take, gidx, best.gidx);
    best.slot = select(take, slot, best.slot);
    best.u = select(take, u, best.u);
    best.v = select(take, v, best.v);
}

// Moeller-Trumbore, no culling; edges inclusive; NaNs reject.  Written without early exits (lanes of a wave rarely
// agree on them); every value is the one the exiting form would have computed.
RT_DEV LaneMask tri_test(V3 v0, V3 e1, V3 e2, V3 o, V3 d, float tnear, float &t, float &u, float &v) {
    V3 pv = cross(d, e2);
    float det = dot(e1, pv);
    float inv = 1.0f / det;
    V3 tv = sub(o, v0);
    u = dot(tv, pv) * inv;
    V3 q = cross(tv, e1);
    v = dot(d, q) * inv;
    t = dot(e2, q) * inv;
    return lanes(det != 0.0f) & lanes(u >= 0.0f) & lanes(u <= 1.0f) & lanes(v >= 0.0f) & lanes(u + v <= 1.0f) & lanes(t > tnear);
}

RT_DEV V3 select_corner(uint32_t s, V3 a, V3 b, V3 c, V3 d) {
    const LaneMask odd = lanes((s & 1u) != 0u), high = lanes((s & 2u) != 0u);
    V3 lo = mk(select(odd, b.x, a.x), select(odd, b.y, a.y), select(odd, b.z, a.z));
    V3 hi = mk(select(odd, d.x, c.x), select(odd, d.y, c.y), select(odd, d.z, c.z));
    return mk(select(high, hi.x, lo.x), select(high, hi.y, lo.y), select(high, hi.z, lo.z));
}

// Corners of triangle `which` (0 or 1) of a record, in the triangle's own vertex order.
RT_DEV void record_corners(F4 r0, F4 r1, F4 r2, uint32_t code, uint32_t which, V3 &v0, V3 &v1, V3 &v2) {
    V3 a = mk(r0.x, r0.y, r0.z), b = mk(r0.w, r1.x, r1.y), c = mk(r1.z, r1.w, r2.x), d = mk(r2.y, r2.z, r2.w);
    if (which == 0u) {
        v0 = a;
        v1 = b;
        v2 = c;
    } else {
        v0 = select_corner(code >> 24, a, b, c, d);
        v1 = select_corner(code >> 26, a, b, c, d);
        v2 = select_corner(code >> 28, a, b, c, d);
    }
}

// Embree's sphere-point formulation: project the centre on the ray, compare the perpendicular distance with the
// radius; Ng = hit - centre = -+td * d - perp.  False when the ray's line misses the sphere.
RT_DEV bool sphere_roots(V3 centre, float radius, V3 o, V3 d, float &t_front, float &t_back, V3 &ng_front, V3 &ng_back) {
    float rd2 = 1.0f / dot(d, d);
    V3 c0 = sub(centre, o);
    float proj = dot(c0, d) * rd2;
    V3 perp = sub(c0, scale(proj, d));
    float l2 = dot(perp, perp);
    float r2s = radius * radius;
    t_front = t_back = 0.0f;
    ng_front = ng_back = mk(0.0f, 0.0f, 0.0f);
    if (!(l2 <= r2s)) return false;
    float td = sqrtf((r2s - l2) * rd2);
    t_front = proj - td;
    t_back = proj + td;
    ng_front = sub(scale(-td, d), perp);
    ng_back = sub(scale(td, d), perp);
    return true;
}

// Tests every primitive of one record.  Closest::slot = 2 * record slot + triangle.
RT_DEV void test_prim(const SceneView &sv, uint32_t slot, V3 o, V3 d, float tnear, Closest &best) {
    const F4 *p = record_at(sv, slot);
    F4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
    keep_whole(r0);
    keep_whole(r1);
    keep_whole(r2);
    keep_whole(r3);
    const uint32_t gidx0 = __builtin_bit_cast(uint32_t, r3.x), gidx1 = __builtin_bit_cast(uint32_t, r3.y);
    const uint32_t code = __builtin_bit_cast(uint32_t, r3.z);
    if (code & RT_SPHERE) {
        float t_front, t_back;
        V3 ng_front, ng_back;
        if (!sphere_roots(mk(r0.x, r0.y, r0.z), r0.w, o, d, t_front, t_back, ng_front, ng_back)) return;
        float t;
        if (tnear <= t_front && t_front <= best.t)
            t = t_front;
        else if (tnear <= t_back && t_back <= best.t)
            t = t_back;
        else
            return;
        take_candidate(t, gidx0, 2u * slot, 0.0f, 0.0f, lanes(true), best);
        return;
    }
    // one triangle after the other: letting the compiler interleave the two tests costs ~20 registers, and the
    // traversal kernel gains more from the extra resident wave than from the overlap
    {
        V3 v0, v1, v2;
        record_corners(r0, r1, r2, code, 0u, v0, v1, v2);
        float t, u, v;
        const LaneMask ok = tri_test(v0, sub(v1, v0), sub(v2, v0), o, d, tnear, t, u, v);
        take_candidate(t, gidx0, 2u * slot, u, v, ok, best);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
    {
        V3 v0, v1, v2;
        record_corners(r0, r1, r2, code, 1u, v0, v1, v2);
        float t, u, v;
        const LaneMask ok = tri_test(v0, sub(v1, v0), sub(v2, v0), o, d, tnear, t, u, v) & lanes(gidx1 != RT_EMPTY);
        take_candidate(t, gidx1, 2u * slot + 1u, u, v, ok, best);
    }
}

// Geometric normal (not normalised) of the accepted candidate, the word holding its geometry id and the global index
// of the first primitive of that geometry.  A sphere's roots are evaluated again: the same operations on the same
// operands give the same bits as during the test, and the front root is the one taken when both equal t.
RT_DEV V3 candidate_normal(const SceneView &sv, uint32_t slot2, V3 o, V3 d, float t, uint32_t &code, uint32_t &geom_first) {
    const F4 *p = record_at(sv, slot2 >> 1);
    F4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
    code = __builtin_bit_cast(uint32_t, r3.z);
    geom_first = __builtin_bit_cast(uint32_t, r3.w);
    if (code & RT_SPHERE) {
        float t_front, t_back;
        V3 ng_front, ng_back;
        sphere_roots(mk(r0.x, r0.y, r0.z), r0.w, o, d, t_front, t_back, ng_front, ng_back);
        return t == t_front ? ng_front : ng_back;
    }
    V3 v0, v1, v2;
    record_corners(r0, r1, r2, code, slot2 & 1u, v0, v1, v2);
    return cross(sub(v1, v0), sub(v2, v0));
}

// Reciprocal that never produces inf, so that 0 * inv is 0 and not NaN in the slab test.  It only feeds the
// conservative box test (whose padding absorbs an error of a few ulp), never a hit, so the device uses the hardware
// approximation (1 ulp) instead of an IEEE division.
RT_DEV float safe_rcp(float d) {
    const float x = fabsf(d) > 1e-20f ? d : copysignf(1e-20f, d);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// Traversal state of one ray: reciprocal direction, the node or leaf to visit next (RT_EMPTY = finished) and the best
// candidate so far.  The stack is separate (`Stack` offers push(uint32_t) / pop() / empty()); the tree's worst-case
// stack need is known at commit time and the kernels provide that much room.
struct Trav {
    float ix, iy, iz;    // safe reciprocals of the direction
    float nox, noy, noz; // -origin * reciprocal, so that a slab distance is one fma
    uint32_t cur;
    Closest best;
};

RT_DEV void trav_begin(const SceneView &sv, V3 o, V3 d, float tfar, Trav &tr) {
    tr.best.t = tfar;
    tr.best.gidx = RT_EMPTY;
    tr.best.slot = 0;
    tr.best.u = tr.best.v = 0.0f;
    tr.ix = safe_rcp(d.x);
    tr.iy = safe_rcp(d.y);
    tr.iz = safe_rcp(d.z);
    tr.nox = -(o.x * tr.ix);
    tr.noy = -(o.y * tr.iy);
    tr.noz = -(o.z * tr.iz);
    tr.cur = sv.n_nodes == 0 ? RT_EMPTY : 0u; // root
}

// One inner-node step (tr.cur must be an inner node): slab-test the four child boxes, continue with the nearest one
// hit and push the others, far to near; with no child hit, pop.  The slab test only has to be conservative (boxes are
// padded at build time), so it uses fused multiply-adds; it never decides a hit.
template <class Stack>
RT_DEV void trav_node_step(const SceneView &sv, float tnear, Trav &tr, Stack &st) {
    const U4 *n = node_at(sv, tr.cur);
    const U4 w0 = n[0], w1 = n[1], w2 = n[2];
    const uint32_t meta = w2.z, first = w2.w;
    // child k as (leaf bit | k): the item index is added once the children are ordered
    const uint32_t ch[4] = {((meta << 15) & RT_LEAF) | 0u, ((meta << 14) & RT_LEAF) | 1u, ((meta << 13) & RT_LEAF) | 2u,
                            ((meta << 12) & RT_LEAF) | 3u};
    const uint32_t last = (meta >> 24) & 3u; // children - 1
    // plane distance = (origin + q * step - o) / d = q * (step / d) + (origin / d - o / d)
    const float sx = __builtin_bit_cast(float, w0.w << 16) * tr.ix;
    const float sy = __builtin_bit_cast(float, w0.w & 0xFFFF0000u) * tr.iy;
    const float sz = __builtin_bit_cast(float, meta << 16) * tr.iz;
    const float bx = __builtin_fmaf(__builtin_bit_cast(float, w0.x), tr.ix, tr.nox);
    const float by = __builtin_fmaf(__builtin_bit_cast(float, w0.y), tr.iy, tr.noy);
    const float bz = __builtin_fmaf(__builtin_bit_cast(float, w0.z), tr.iz, tr.noz);
    // the plane a ray meets first on an axis is the low one when it travels in the positive direction
    const LaneMask nx = lanes(tr.ix < 0.0f), ny = lanes(tr.iy < 0.0f), nz = lanes(tr.iz < 0.0f);
    const uint32_t qnx = select(nx, w1.w, w1.x), qfx = select(nx, w1.x, w1.w);
    const uint32_t qny = select(ny, w2.x, w1.y), qfy = select(ny, w1.y, w2.x);
    const uint32_t qnz = select(nz, w2.y, w1.z), qfz = select(nz, w1.z, w2.y);
    const LaneMask exists[4] = {lanes(true), lanes(last >= 1u), lanes(last >= 2u), lanes(last >= 3u)};
    const float tfar = tr.best.t;
    float tn[4];
    LaneMask hit[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sh = 8 * k;
        const float t0x = __builtin_fmaf((float)((qnx >> sh) & 255u), sx, bx);
        const float t1x = __builtin_fmaf((float)((qfx >> sh) & 255u), sx, bx);
        const float t0y = __builtin_fmaf((float)((qny >> sh) & 255u), sy, by);
        const float t1y = __builtin_fmaf((float)((qfy >> sh) & 255u), sy, by);
        const float t0z = __builtin_fmaf((float)((qnz >> sh) & 255u), sz, bz);
        const float t1z = __builtin_fmaf((float)((qfz >> sh) & 255u), sz, bz);
        const float tmin = fmaxf(fmaxf(t0x, t0y), fmaxf(t0z, tnear));
        const float tmax = fminf(fminf(t1x, t1y), fminf(t1z, tfar));
        hit[k] = exists[k] & lanes(tmin <= tmax * 1.0000004f);
        tn[k] = select(hit[k], tmin, INFINITY);
    }
    // 4-element sorting network on (entry distance, child), nearest first; misses carry +inf and RT_EMPTY
    float a0 = tn[0], a1 = tn[1], a2 = tn[2], a3 = tn[3];
    uint32_t i0 = select(hit[0], ch[0], RT_EMPTY), i1 = select(hit[1], ch[1], RT_EMPTY);
    uint32_t i2 = select(hit[2], ch[2], RT_EMPTY), i3 = select(hit[3], ch[3], RT_EMPTY);
#define RT_CSWAP(ta, ia, tb, ib)               \
    {                                          \
        const LaneMask sw = lanes(tb < ta);    \
        const float tlo = select(sw, tb, ta);  \
        const float thi = select(sw, ta, tb);  \
        const uint32_t ilo = select(sw, ib, ia); \
        const uint32_t ihi = select(sw, ia, ib); \
        ta = tlo;                              \
        tb = thi;                              \
        ia = ilo;                              \
        ib = ihi;                              \
    }
    RT_CSWAP(a0, i0, a1, i1)
    RT_CSWAP(a2, i2, a3, i3)
    RT_CSWAP(a0, i0, a2, i2)
    RT_CSWAP(a1, i1, a3, i3)
    RT_CSWAP(a1, i1, a2, i2)
#undef RT_CSWAP
    if (i0 == RT_EMPTY) {
        tr.cur = st.empty() ? RT_EMPTY : st.pop();
    } else {
        if (i3 != RT_EMPTY) st.push(first + i3);
        if (i2 != RT_EMPTY) st.push(first + i2);
        if (i1 != RT_EMPTY) st.push(first + i1);
        tr.cur = first + i0;
    }
}

// One leaf step (tr.cur must be a leaf reference other than RT_EMPTY): test the leaf's first primitive, then continue
// with the rest of the leaf or pop.
template <class Stack>
RT_DEV void trav_leaf_step(const SceneView &sv, V3 o, V3 d, float tnear, Trav &tr, Stack &st) {
    const uint32_t first = tr.cur & 0x0FFFFFFFu;
    const uint32_t more = (tr.cur >> 28) & 7u; // primitives after this one
    test_prim(sv, first, o, d, tnear, tr.best);
    if (more != 0u)
        tr.cur = RT_LEAF | ((more - 1u) << 28) | (first + 1u);
    else
        tr.cur = st.empty() ? RT_EMPTY : st.pop();
}

// Closest hit of o + t d over (tnear, tfar] in the 4-wide BVH, one lane on its own.  Visit counters (may be null) count
// node and primitive fetches.
template <class Stack>
RT_DEV void closest_hit(const SceneView &sv, V3 o, V3 d, float tnear, float tfar, Stack &st, Closest &best,
                        uint32_t *node_visits, uint32_t *prim_tests) {
    Trav tr;
    trav_begin(sv, o, d, tfar, tr);
    while (tr.cur != RT_EMPTY) {
        if (!(tr.cur & RT_LEAF)) {
            if (node_visits) ++*node_visits;
            trav_node_step(sv, tnear, tr, st);
        } else {
            if (prim_tests) ++*prim_tests;
            trav_leaf_step(sv, o, d, tnear, tr, st);
        }
    }
    best = tr.best;
}

struct HitInfo {
    bool hit;
    float t;
    V3 pos;
    V3 n;
    float u, v;
    uint32_t geom_id, prim_id, mat_id;
};

// embree.rs:115-137: position recomputed as org + t * dir, normal = normalize_or_zero(Ng), material by geometry.
RT_DEV void finish_hit(const SceneView &sv, V3 o, V3 d, const Closest &best, HitInfo &h) {
    if (best.gidx == RT_EMPTY) {
        h.hit = false;
        h.t = INFINITY;
        h.pos = mk(0.0f, 0.0f, 0.0f);
        h.n = mk(0.0f, 0.0f, 0.0f);
        h.u = h.v = 0.0f;
        h.geom_id = h.prim_id = RT_EMPTY;
        h.mat_id = 0;
        return;
    }
    uint32_t code, geom_first;
    const V3 ng = candidate_normal(sv, best.slot, o, d, best.t, code, geom_first);
    h.hit = true;
    h.t = best.t;
    h.pos = mk(o.x + best.t * d.x, o.y + best.t * d.y, o.z + best.t * d.z);
    h.n = normalize_or_zero(ng);
    h.u = best.u;
    h.v = best.v;
    h.geom_id = code & RT_GEOM_MASK;
    h.prim_id = best.gidx - geom_first;
    h.mat_id = h.geom_id < sv.n_geoms ? sv.geom_material[h.geom_id] : 0u;
    if (h.mat_id >= sv.n_materials) h.mat_id = 0u;
}

// ---------------------------------------------------------------------------------------------- shading step

// One vertex of the front-to-back path estimator.  On entry beta is the throughput carried to this vertex.
// Returns true when the path continues (org/dir/beta updated); otherwise `radiance` holds the path's final colour.
// first-hit outputs (normal, position, albedo, z) are meaningful when depth == 0.
struct VertexOut {
    V3 normal, position, albedo;
    float z;
    float t_add; // distance to add to the path's ray_depth
    V3 radiance;
};

RT_DEV bool shade_vertex(const SceneView &sv, const HitInfo &h, uint32_t integrator, uint64_t seed, uint32_t x, uint32_t y,
                         uint32_t sample, uint32_t depth, V3 &org, V3 &dir, V3 &beta, VertexOut &out) {
    out.normal = mk(0.0f, 0.0f, 0.0f);
    out.position = mk(0.0f, 0.0f, 0.0f);
    out.albedo = mk(0.0f, 0.0f, 0.0f);
    out.z = 0.0f;
    out.t_add = 0.0f;
    if (!h.hit) {
        out.radiance = mul(beta, mk(0.5f, 0.3f, 1.0f)); // integrators/mod.rs:19
        return false;
    }
    const Material m = sv.materials[h.mat_id];
    float u[4];
    uniform4(seed, x, y, sample, 1u + depth, u);
    V3 wo = neg(dir);
    V3 wi, f, weight_scale_fcos;
    if (integrator == 0u) {
        BxdfSample sm;
        if (!bsdf_sample_f(m, h.n, wo, u[0], u[1], u[2], sm)) {
            sm.wi = mk(0.0f, 0.0f, 0.0f);
            sm.f = mk(0.0f, 0.0f, 0.0f);
            sm.pdf = 1.0f;
        }
        wi = sm.wi;
        f = sm.f;
        V3 fcos = scale(fabsf(dot(h.n, wi)), f);
        out.normal = h.n;
        out.position = h.pos;
        out.albedo = f;
        out.z = h.t;
        out.t_add = h.t;
        if (!(fabsf(max_element(fcos)) != 0.0f)) {
            out.radiance = mul(beta, material_le(m));
            return false;
        }
        weight_scale_fcos = scale(1.0f / sm.pdf, fcos); // pathtracing.rs:60
    } else {
        wi = uniform_unit_sphere3(u[0], u[1]);
        f = bsdf_f(m, h.n, wo, wi);
        V3 fcos = scale(fabsf(dot(h.n, wi)), f);
        out.normal = h.n;
        out.position = h.pos;
        out.albedo = f;
        out.z = h.t;
        out.t_add = (depth == 0u) ? h.t : 0.0f; // randomwalk.rs:65: ray_depth is the first hit's t only
        if (!(fabsf(max_element(fcos)) != 0.0f)) {
            out.radiance = mul(beta, material_le(m));
            return false;
        }
        weight_scale_fcos = scale(RT_FRAC_1_PI / 4.0f, fcos); // randomwalk.rs:54
    }
    beta = mul(beta, weight_scale_fcos);
    org = h.pos;
    dir = wi;
    return true;
}

} // namespace rtd
