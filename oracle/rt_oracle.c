/*
 * rt_oracle.c -- CPU restatement of the philae-ael/Raytracing render path.
 * TEST INFRASTRUCTURE ONLY; see rt_oracle.h for scope, citations and parity status.
 *
 * All arithmetic is f32 in the written order.  Build with -ffp-contract=off and without -ffast-math.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ vectors */

typedef struct {
    float x, y, z;
} v3;

static inline v3 V(float x, float y, float z) {
    v3 r = {x, y, z};
    return r;
}
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* glam Vec3::dot: (x*x' + y*y') + z*z' */
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* glam Vec3::cross */
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
/* glam Vec3::normalize: self * (1 / sqrt(dot)) */
static inline v3 vnormalize(v3 a) {
    float rcp = 1.0f / sqrtf(vdot(a, a));
    return V(a.x * rcp, a.y * rcp, a.z * rcp);
}
/* glam Vec3::normalize_or_zero */
static inline v3 vnormalize_or_zero(v3 a) {
    float rcp = 1.0f / sqrtf(vdot(a, a));
    if (isfinite(rcp) && rcp > 0.0f) return V(a.x * rcp, a.y * rcp, a.z * rcp);
    return V(0.0f, 0.0f, 0.0f);
}
static inline float vmax_element(v3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
static inline v3 v3_from(const float *p) { return V(p[0], p[1], p[2]); }
static inline void v3_to(v3 a, float *p) {
    p[0] = a.x;
    p[1] = a.y;
    p[2] = a.z;
}

#define ORC_PI 3.14159265358979323846f
#define ORC_FRAC_1_PI 0.318309886183790671538f
#define ORC_INF INFINITY

/* ------------------------------------------------------------------------------------------------ RNG */

/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
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

/* Four uniforms in [0,1) for (seed, x, y, sample, stream).  The float mapping is rand 0.8.5's Uniform<f32>(0,1):
 * the top 23 bits as k * 2^-23.  stream 0 = camera sample, stream 1 + d = bounce d. */
void orc_uniform4(uint64_t seed, uint32_t x, uint32_t y, uint32_t sample, uint32_t stream, float out[4]) {
    uint32_t ctr[4] = {x, y, sample, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t r[4];
    orc_philox4x32_10(ctr, key, r);
    for (int i = 0; i < 4; ++i) out[i] = (float)(r[i] >> 9) * 1.1920928955078125e-07f;
}

/* sin(2 pi u), cos(2 pi u) for u in [0,1): nearest quarter turn removed exactly, then Taylor polynomials in
 * x = 2 pi r, |x| <= pi/4, evaluated by Horner in plain multiply/add (error < 2e-9 before rounding). */
void orc_sincos_2pi(float u, float *s, float *c) {
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
        *s = sp;
        *c = cp;
    } else if (q == 1) {
        *s = cp;
        *c = -sp;
    } else if (q == 2) {
        *s = -sp;
        *c = -cp;
    } else {
        *s = -cp;
        *c = sp;
    }
}

/* ------------------------------------------------------------------------------------------------ frame */

typedef struct {
    v3 c0, c1, c2;
} frame3;

/* math/transform.rs:63-86 */
static frame3 frame_new(v3 n) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    frame3 f;
    f.c0 = V(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    f.c1 = V(b, sign + n.y * n.y * a, -n.y);
    f.c2 = n;
    return f;
}
/* transform.rs:88-90 (Mat3 transpose * v) */
static v3 frame_to_local(const frame3 *f, v3 g) { return V(vdot(f->c0, g), vdot(f->c1, g), vdot(f->c2, g)); }
/* transform.rs:92-94 (Mat3 * v = c0*x + c1*y + c2*z) */
static v3 frame_from_local(const frame3 *f, v3 l) {
    return V((f->c0.x * l.x + f->c1.x * l.y) + f->c2.x * l.z, (f->c0.y * l.x + f->c1.y * l.y) + f->c2.y * l.z,
             (f->c0.z * l.x + f->c1.z * l.y) + f->c2.z * l.z);
}

void orc_frame(const float n[3], float cols[9]) {
    frame3 f = frame_new(v3_from(n));
    v3_to(f.c0, cols);
    v3_to(f.c1, cols + 3);
    v3_to(f.c2, cols + 6);
}

/* ------------------------------------------------------------------------------------------------ sampling */

/* distributions.rs:86-95 */
static void ball2(float u0, float u1, float *px, float *py) {
    float s, c;
    orc_sincos_2pi(u0, &s, &c);
    float r = sqrtf(u1);
    *px = r * c;
    *py = r * s;
}

/* distributions.rs:176-188 */
static v3 cosine_hemisphere(float u0, float u1) {
    float px, py;
    ball2(u0, u1, &px, &py);
    float z = sqrtf(1.0f - px * px - py * py);
    return V(px, py, z);
}

/* distributions.rs:120-130; theta = pi * u1 is evaluated as 2 pi (u1 / 2) */
static v3 uniform_unit_sphere3(float u0, float u1) {
    float sphi, cphi, stheta, ctheta;
    orc_sincos_2pi(u0, &sphi, &cphi);
    orc_sincos_2pi(u1 * 0.5f, &stheta, &ctheta);
    return V(cphi * stheta, sphi * stheta, ctheta);
}

/* math/float.rs:26-28 */
static inline float lerp_ext(float self, float from, float to) { return self * to + (1.0f - self) * from; }

/* vec.rs:16-18 */
static inline int same_hemisphere(v3 a, v3 b) { return a.z * b.z > 0.0f; }
/* vec.rs:27-29 */
static inline v3 reflect(v3 w, v3 n) { return vadd(vneg(w), vscale(2.0f * vdot(w, n), n)); }

/* vec.rs:33-52 */
static int refract(v3 w, v3 normal, float ior, v3 *out, float *ior_out) {
    float cosi = vdot(w, normal);
    if (!(cosi >= 0.0f)) {
        normal = vneg(normal);
        cosi = -cosi;
        ior = 1.0f / ior;
    }
    float cost2 = 1.0f - fmaxf(0.0f, 1.0f - cosi * cosi) / ior / ior;
    if (cost2 <= 0.0f) return 0;
    float cost = sqrtf(cost2);
    v3 a = V(-w.x / ior, -w.y / ior, -w.z / ior);
    *out = vadd(a, vscale(cosi / ior - cost, normal));
    *ior_out = ior;
    return 1;
}

int orc_refract(const float w[3], const float n[3], float ior, float out[3], float *ior_out) {
    v3 o;
    float io = 0.0f;
    int ok = refract(v3_from(w), v3_from(n), ior, &o, &io);
    if (ok) {
        v3_to(o, out);
        *ior_out = io;
    }
    return ok;
}

/* material/mod.rs:139-154 */
float orc_fresnel_dielectric(float cosi, float ior) {
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

/* distributions.rs:203-265, formulas kept exactly as written there */
static inline int tr_is_smooth(float alpha) { return alpha < 1e-8f; }
static float tr_d(float alpha, v3 wm) {
    float cos2 = wm.z * wm.z;
    float sin2 = 1.0f - cos2;
    float tan2 = sin2 / cos2;
    float at = alpha + tan2;
    return 1.0f / (ORC_PI * (cos2 * cos2) * (at * at));
}
static float tr_lambda(float alpha, v3 w) {
    float cos2 = w.z * w.z;
    float sin2 = 1.0f - cos2;
    float tan2 = sin2 / cos2;
    return sqrtf(1.0f + (alpha * alpha) * tan2) / 2.0f;
}
static float tr_g(float alpha, v3 w0, v3 w1) { return 1.0f / (1.0f + tr_lambda(alpha, w0) + tr_lambda(alpha, w1)); }
static float tr_g1(float alpha, v3 w) { return 1.0f / (1.0f + tr_lambda(alpha, w)); }
static float tr_dw(float alpha, v3 w, v3 wm) {
    return tr_g1(alpha, w) / fabsf(w.z) * tr_d(alpha, wm) * fabsf(vdot(w, wm));
}
static v3 tr_sample_wm(float alpha, v3 w, float u0, float u1) {
    w = vnormalize(V(alpha * w.x, alpha * w.y, w.z));
    if (!(w.z >= 0.0f)) w = vneg(w);
    v3 t1 = (w.z < 0.9999f) ? vnormalize(vcross(V(0.0f, 0.0f, 1.0f), w)) : V(1.0f, 0.0f, 0.0f);
    v3 t2 = vcross(w, t1);
    float px, py;
    ball2(u0, u1, &px, &py);
    float h = sqrtf(1.0f - px * px);
    py = lerp_ext((1.0f + w.z) / 2.0f, h, py);
    float pz = sqrtf(fmaxf(0.0f, 1.0f - (px * px + py * py)));
    v3 nh = vadd(vadd(vscale(px, t1), vscale(py, t2)), vscale(pz, w));
    return vnormalize(V(alpha * nh.x, alpha * nh.y, fmaxf(1e-6f, nh.z)));
}

/* ------------------------------------------------------------------------------------------------ BxDFs */

typedef struct {
    v3 wi;
    v3 f;
    float pdf;
} bxdf_sample;

static inline v3 white(float s) { return V(s * 1.0f, s * 1.0f, s * 1.0f); }

/* material/mod.rs:119-130 */
static int diffuse_sample_f(const orc_material *m, v3 wo, float u0, float u1, bxdf_sample *o) {
    v3 wi = cosine_hemisphere(u0, u1);
    wi.z = copysignf(wi.z, wo.z);
    o->wi = wi;
    o->f = V(ORC_FRAC_1_PI * m->rgb[0], ORC_FRAC_1_PI * m->rgb[1], ORC_FRAC_1_PI * m->rgb[2]);
    o->pdf = fabsf(wi.z) * ORC_FRAC_1_PI;
    return 1;
}

/* material/mod.rs:244-312 */
static int dielectric_sample_f(const orc_material *m, v3 wo, float u0, float u1, float u2, bxdf_sample *o) {
    float alpha = m->roughness;
    if (m->ior == 1.0f || tr_is_smooth(alpha)) {
        float r = orc_fresnel_dielectric(wo.z, m->ior);
        float t = 1.0f - r;
        if (u2 <= r / (r + t)) {
            v3 wi = V(-wo.x, -wo.y, wo.z);
            o->wi = wi;
            o->f = white(r / fabsf(wi.z));
            o->pdf = r / (r + t);
            return 1;
        }
        v3 wi;
        float ior2;
        if (!refract(wo, V(0.0f, 0.0f, 1.0f), m->ior, &wi, &ior2)) return 0;
        o->wi = wi;
        o->f = white(t / fabsf(wi.z));
        o->pdf = t / (r + t);
        return 1;
    }
    v3 wm = tr_sample_wm(alpha, wo, u0, u1);
    float r = orc_fresnel_dielectric(vdot(wo, wm), m->ior);
    float t = 1.0f - r;
    if (u2 < r / (r + t)) {
        v3 wi = reflect(wo, wm);
        if (!same_hemisphere(wo, wi)) return 0;
        float f = tr_d(alpha, wm) * tr_g(alpha, wo, wi) * r / (4.0f * wi.z * wo.z);
        float pdf = tr_dw(alpha, wo, wm) / (4.0f * fabsf(vdot(wo, wm))) * r / (r + t);
        o->wi = wi;
        o->f = white(f);
        o->pdf = pdf;
        return 1;
    }
    v3 wi;
    float ior2;
    if (!refract(wo, wm, m->ior, &wi, &ior2)) return 0;
    if (same_hemisphere(wi, wo) || wi.z == 0.0f) return 0;
    float s = vdot(wi, wm) + vdot(wo, wm) / ior2;
    float denom = s * s;
    float dwm_dwi = fabsf(vdot(wi, wm)) / denom;
    float f = t * tr_d(alpha, wm) * tr_g(alpha, wo, wi) * fabsf(vdot(wo, wm) * vdot(wi, wm) / (wi.z * wo.z * denom));
    float pdf = tr_dw(alpha, wo, wm) * dwm_dwi * t / (r + t);
    o->wi = wi;
    o->f = white(f);
    o->pdf = pdf;
    return 1;
}

/* material/mod.rs:339-366 */
static int thin_dielectric_sample_f(const orc_material *m, v3 wo, float u2, bxdf_sample *o) {
    float r = orc_fresnel_dielectric(wo.z, m->ior);
    float t = 1.0f - r;
    if (r >= 1.0f) {
        r = 1.0f;
        t = 0.0f;
    } else {
        r = r + t * t * r / (1.0f - r * r);
        t = 1.0f - r;
    }
    if (u2 <= r / (r + t)) {
        v3 wi = V(-wo.x, -wo.y, wo.z);
        o->wi = wi;
        o->f = white(r / fabsf(wi.z));
        o->pdf = r / (r + t);
    } else {
        v3 wi = vneg(wo);
        o->wi = wi;
        o->f = white(t / fabsf(wi.z));
        o->pdf = t / (r + t);
    }
    return 1;
}

/* material/mod.rs:167-207 (only the random-walk integrator evaluates f) */
static v3 dielectric_f(const orc_material *m, v3 wo, v3 wi) {
    float alpha = m->roughness;
    v3 black = V(0.0f, 0.0f, 0.0f);
    if (m->ior == 1.0f || tr_is_smooth(alpha)) return black;
    float cosi = wi.z, coso = wo.z;
    int is_reflect = cosi * coso > 0.0f;
    float ior = (is_reflect || coso > 0.0f) ? m->ior : 1.0f / m->ior;
    v3 wm = vadd(vscale(ior, wi), wo);
    if (cosi == 0.0f || coso == 0.0f || vdot(wm, wm) == 0.0f) return black;
    /* f32::signum: +1 for +0, -1 for -0, NaN for NaN */
    float sg = isnan(wm.z) ? wm.z : copysignf(1.0f, wm.z);
    wm = vscale(sg, vnormalize(wm));
    if (vdot(wi, wm) * cosi < 0.0f || vdot(wo, wm) * coso < 0.0f) return black;
    float r = orc_fresnel_dielectric(vdot(wo, wm), ior);
    float t = 1.0f - r;
    if (is_reflect) return white(tr_d(alpha, wm) * tr_g(alpha, wo, wi) * r / fabsf(4.0f * cosi * coso));
    float s = vdot(wi, wm) + vdot(wo, wm) / ior;
    float denom = s * s * cosi * coso;
    return white(tr_d(alpha, wm) * t * tr_g(alpha, wo, wi) * fabsf(vdot(wi, wm) * vdot(wo, wm) / denom));
}

/* BSDF::sample_f, material/mod.rs:76-86 */
static int bsdf_sample_f(const orc_material *m, v3 n, v3 wo, float u0, float u1, float u2, bxdf_sample *o) {
    frame3 fr = frame_new(n);
    v3 wo_local = frame_to_local(&fr, wo);
    if (wo_local.z == 0.0f) return 0;
    int ok;
    switch (m->kind) {
    case ORC_MAT_DIFFUSE: ok = diffuse_sample_f(m, wo_local, u0, u1, o); break;
    case ORC_MAT_DIELECTRIC: ok = dielectric_sample_f(m, wo_local, u0, u1, u2, o); break;
    case ORC_MAT_THIN_DIELECTRIC: ok = thin_dielectric_sample_f(m, wo_local, u2, o); break;
    default: ok = 0; break; /* EmitBxDF::sample_f -> None, material/mod.rs:387-389 */
    }
    if (!ok) return 0;
    o->wi = frame_from_local(&fr, o->wi);
    return 1;
}

/* BSDF::f, material/mod.rs:88-91 */
static v3 bsdf_f(const orc_material *m, v3 n, v3 wo, v3 wi) {
    frame3 fr = frame_new(n);
    v3 wol = frame_to_local(&fr, wo);
    v3 wil = frame_to_local(&fr, wi);
    switch (m->kind) {
    case ORC_MAT_DIFFUSE: /* material/mod.rs:108-113 */
        if (!same_hemisphere(wol, wil)) return V(0.0f, 0.0f, 0.0f);
        return V(ORC_FRAC_1_PI * m->rgb[0], ORC_FRAC_1_PI * m->rgb[1], ORC_FRAC_1_PI * m->rgb[2]);
    case ORC_MAT_DIELECTRIC: return dielectric_f(m, wol, wil);
    default: return V(0.0f, 0.0f, 0.0f);
    }
}

static inline v3 material_le(const orc_material *m) {
    if (m->kind == ORC_MAT_EMIT) return v3_from(m->rgb);
    return V(0.0f, 0.0f, 0.0f);
}

int orc_bsdf_sample(const orc_material *m, const float n[3], const float wo[3], const float u[3], float wi[3],
                    float f[3], float *pdf) {
    bxdf_sample o;
    if (!bsdf_sample_f(m, v3_from(n), v3_from(wo), u[0], u[1], u[2], &o)) return 0;
    v3_to(o.wi, wi);
    v3_to(o.f, f);
    *pdf = o.pdf;
    return 1;
}

void orc_bsdf_f(const orc_material *m, const float n[3], const float wo[3], const float wi[3], float f[3]) {
    v3_to(bsdf_f(m, v3_from(n), v3_from(wo), v3_from(wi)), f);
}

/* ------------------------------------------------------------------------------------------------ primitives */

/* Moeller-Trumbore, no culling.  Accepts u >= 0, v >= 0, u + v <= 1 (edges inclusive); NaNs reject. */
static inline int ray_triangle(v3 o, v3 d, v3 v0, v3 v1, v3 v2, float *t, float *u, float *v) {
    v3 e1 = vsub(v1, v0);
    v3 e2 = vsub(v2, v0);
    v3 p = vcross(d, e2);
    float det = vdot(e1, p);
    if (!(det != 0.0f)) return 0;
    float inv = 1.0f / det;
    v3 tv = vsub(o, v0);
    float uu = vdot(tv, p) * inv;
    if (!(uu >= 0.0f && uu <= 1.0f)) return 0;
    v3 q = vcross(tv, e1);
    float vv = vdot(d, q) * inv;
    if (!(vv >= 0.0f && uu + vv <= 1.0f)) return 0;
    *t = vdot(e2, q) * inv;
    *u = uu;
    *v = vv;
    return 1;
}

/* Embree 4 sphere-point intersector (kernels/geometry/sphere_intersector.h, third-party, restated): project the
 * centre on the ray, compare the perpendicular distance with the radius; Ng = hit - centre = -+td*d - perp. */
static inline int ray_sphere(v3 o, v3 d, float tnear, float tfar, v3 c, float radius, float *t, v3 *ng) {
    float rd2 = 1.0f / vdot(d, d);
    v3 c0 = vsub(c, o);
    float proj = vdot(c0, d) * rd2;
    v3 perp = vsub(c0, vscale(proj, d));
    float l2 = vdot(perp, perp);
    float r2 = radius * radius;
    if (!(l2 <= r2)) return 0;
    float td = sqrtf((r2 - l2) * rd2);
    float t_front = proj - td;
    float t_back = proj + td;
    if (tnear <= t_front && t_front <= tfar) {
        *t = t_front;
        *ng = vsub(vscale(-td, d), perp);
        return 1;
    }
    if (tnear <= t_back && t_back <= tfar) {
        *t = t_back;
        *ng = vsub(vscale(td, d), perp);
        return 1;
    }
    return 0;
}

int orc_ray_triangle(const orc_ray *ray, const float v0[3], const float v1[3], const float v2[3], float *t, float *u,
                     float *v) {
    float tt, uu, vv;
    if (!ray_triangle(v3_from(ray->org), v3_from(ray->dir), v3_from(v0), v3_from(v1), v3_from(v2), &tt, &uu, &vv))
        return 0;
    if (!(tt > ray->tnear && tt <= ray->tfar)) return 0;
    *t = tt;
    *u = uu;
    *v = vv;
    return 1;
}

int orc_ray_sphere(const orc_ray *ray, const float center[3], float radius, float *t, float ng[3]) {
    v3 n;
    if (!ray_sphere(v3_from(ray->org), v3_from(ray->dir), ray->tnear, ray->tfar, v3_from(center), radius, t, &n))
        return 0;
    v3_to(n, ng);
    return 1;
}

/* Slab test of math/bounds.rs:61-91 made conservative: an axis on which the ray is parallel to and exactly on a
 * slab plane (0 * inf = NaN) does not constrain the interval. */
static inline int slab(v3 lo, v3 hi, v3 o, v3 inv, float tnear, float tfar, float *t0, float *t1) {
    float tmin = tnear, tmax = tfar;
    const float los[3] = {lo.x, lo.y, lo.z}, his[3] = {hi.x, hi.y, hi.z};
    const float os[3] = {o.x, o.y, o.z}, is[3] = {inv.x, inv.y, inv.z};
    for (int a = 0; a < 3; ++a) {
        float ta = (los[a] - os[a]) * is[a];
        float tb = (his[a] - os[a]) * is[a];
        if (isnan(ta) || isnan(tb)) continue;
        float tn = ta < tb ? ta : tb;
        float tf = ta < tb ? tb : ta;
        if (tn > tmin) tmin = tn;
        if (tf < tmax) tmax = tf;
    }
    *t0 = tmin;
    *t1 = tmax;
    return tmin <= tmax * 1.0000004f;
}

int orc_slab_test(const float lo[3], const float hi[3], const orc_ray *ray, float *t0, float *t1) {
    v3 inv = V(1.0f / ray->dir[0], 1.0f / ray->dir[1], 1.0f / ray->dir[2]);
    return slab(v3_from(lo), v3_from(hi), v3_from(ray->org), inv, ray->tnear, ray->tfar, t0, t1);
}

/* ------------------------------------------------------------------------------------------------ scene */

typedef struct {
    v3 v0, v1, v2; /* sphere: v0 = centre, v1.x = radius */
    uint32_t geom_id, prim_id, mat_id, is_sphere;
} prim_t;

typedef struct {
    v3 lo, hi;
    uint32_t left;  /* inner: index of left child (right = left + 1); leaf: first prim in order[] */
    uint32_t count; /* 0 = inner */
} bvh_node;

struct orc_scene {
    orc_material *materials;
    size_t n_materials, cap_materials;
    float *lights;
    size_t n_lights, cap_lights;
    prim_t *prims; /* global order = insertion order = tie-break order */
    size_t n_prims, cap_prims;
    uint32_t n_geoms;
    uint32_t *geom_material; /* embree.rs:53-61 geometry_material map */
    size_t cap_geoms;
    /* BVH */
    bvh_node *nodes;
    size_t n_nodes;
    uint32_t *order;
    int committed;
    v3 lo, hi;
};

orc_scene *orc_scene_new(void) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    /* embree.rs:41-46: material 0 is the black "Sky" diffuse */
    orc_material sky;
    memset(&sky, 0, sizeof sky);
    sky.kind = ORC_MAT_DIFFUSE;
    orc_scene_insert_material(s, &sky);
    return s;
}

void orc_scene_free(orc_scene *s) {
    if (!s) return;
    free(s->materials);
    free(s->lights);
    free(s->prims);
    free(s->geom_material);
    free(s->nodes);
    free(s->order);
    free(s);
}

uint32_t orc_scene_insert_material(orc_scene *s, const orc_material *m) {
    if (s->n_materials == s->cap_materials) {
        s->cap_materials = s->cap_materials ? 2 * s->cap_materials : 16;
        s->materials = (orc_material *)realloc(s->materials, s->cap_materials * sizeof(orc_material));
    }
    s->materials[s->n_materials] = *m;
    return (uint32_t)s->n_materials++;
}

void orc_scene_insert_light(orc_scene *s, const float pos[3]) {
    if (s->n_lights == s->cap_lights) {
        s->cap_lights = s->cap_lights ? 2 * s->cap_lights : 8;
        s->lights = (float *)realloc(s->lights, 3 * s->cap_lights * sizeof(float));
    }
    memcpy(s->lights + 3 * s->n_lights++, pos, 3 * sizeof(float));
}

static void reserve_prims(orc_scene *s, size_t extra) {
    if (s->n_prims + extra > s->cap_prims) {
        size_t cap = s->cap_prims ? s->cap_prims : 64;
        while (cap < s->n_prims + extra) cap *= 2;
        s->prims = (prim_t *)realloc(s->prims, cap * sizeof(prim_t));
        s->cap_prims = cap;
    }
}

static uint32_t new_geom(orc_scene *s, uint32_t material) {
    if (s->n_geoms == s->cap_geoms) {
        s->cap_geoms = s->cap_geoms ? 2 * s->cap_geoms : 16;
        s->geom_material = (uint32_t *)realloc(s->geom_material, s->cap_geoms * sizeof(uint32_t));
    }
    s->geom_material[s->n_geoms] = material;
    return s->n_geoms++;
}

uint32_t orc_scene_insert_mesh(orc_scene *s, uint32_t material, const float *vertices, size_t n_vertices,
                               const uint32_t *indices, size_t n_triangles) {
    uint32_t g = new_geom(s, material);
    reserve_prims(s, n_triangles);
    for (size_t i = 0; i < n_triangles; ++i) {
        prim_t *p = &s->prims[s->n_prims++];
        uint32_t a = indices[3 * i], b = indices[3 * i + 1], c = indices[3 * i + 2];
        if (a >= n_vertices || b >= n_vertices || c >= n_vertices) a = b = c = 0;
        p->v0 = v3_from(vertices + 3 * a);
        p->v1 = v3_from(vertices + 3 * b);
        p->v2 = v3_from(vertices + 3 * c);
        p->geom_id = g;
        p->prim_id = (uint32_t)i;
        p->mat_id = material;
        p->is_sphere = 0;
    }
    s->committed = 0;
    return g;
}

uint32_t orc_scene_insert_sphere(orc_scene *s, uint32_t material, const float center[3], float radius) {
    uint32_t g = new_geom(s, material);
    reserve_prims(s, 1);
    prim_t *p = &s->prims[s->n_prims++];
    p->v0 = v3_from(center);
    p->v1 = V(radius, 0.0f, 0.0f);
    p->v2 = V(0.0f, 0.0f, 0.0f);
    p->geom_id = g;
    p->prim_id = 0;
    p->mat_id = material;
    p->is_sphere = 1;
    s->committed = 0;
    return g;
}

size_t orc_scene_prim_count(const orc_scene *s) { return s->n_prims; }

static inline float fmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
static inline float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

static void prim_bounds(const prim_t *p, v3 *lo, v3 *hi) {
    if (p->is_sphere) {
        float r = p->v1.x;
        *lo = V(p->v0.x - r, p->v0.y - r, p->v0.z - r);
        *hi = V(p->v0.x + r, p->v0.y + r, p->v0.z + r);
    } else {
        *lo = V(fmin3(p->v0.x, p->v1.x, p->v2.x), fmin3(p->v0.y, p->v1.y, p->v2.y), fmin3(p->v0.z, p->v1.z, p->v2.z));
        *hi = V(fmax3(p->v0.x, p->v1.x, p->v2.x), fmax3(p->v0.y, p->v1.y, p->v2.y), fmax3(p->v0.z, p->v1.z, p->v2.z));
    }
}

/* Boxes used for culling are padded so that a hit the f32 primitive test accepts is never culled. */
static void pad_bounds(v3 *lo, v3 *hi) {
    float m = fmaxf(fmax3(fabsf(lo->x), fabsf(lo->y), fabsf(lo->z)), fmax3(fabsf(hi->x), fabsf(hi->y), fabsf(hi->z)));
    float e = 1e-5f * m + 1e-30f;
    *lo = V(lo->x - e, lo->y - e, lo->z - e);
    *hi = V(hi->x + e, hi->y + e, hi->z + e);
}

typedef struct {
    orc_scene *s;
    v3 *plo, *phi, *cen;
} build_ctx;

static inline float half_area(v3 lo, v3 hi) {
    float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return dx * dy + dy * dz + dz * dx;
}
static inline float axis_of(v3 a, int ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }

#define ORC_BINS 16
#define ORC_LEAF_MAX 4

static void build_node(build_ctx *c, uint32_t node_idx, uint32_t begin, uint32_t end) {
    orc_scene *s = c->s;
    bvh_node *n = &s->nodes[node_idx];
    v3 lo = V(ORC_INF, ORC_INF, ORC_INF), hi = V(-ORC_INF, -ORC_INF, -ORC_INF);
    v3 clo = lo, chi = hi;
    for (uint32_t i = begin; i < end; ++i) {
        uint32_t p = s->order[i];
        lo = V(fminf(lo.x, c->plo[p].x), fminf(lo.y, c->plo[p].y), fminf(lo.z, c->plo[p].z));
        hi = V(fmaxf(hi.x, c->phi[p].x), fmaxf(hi.y, c->phi[p].y), fmaxf(hi.z, c->phi[p].z));
        clo = V(fminf(clo.x, c->cen[p].x), fminf(clo.y, c->cen[p].y), fminf(clo.z, c->cen[p].z));
        chi = V(fmaxf(chi.x, c->cen[p].x), fmaxf(chi.y, c->cen[p].y), fmaxf(chi.z, c->cen[p].z));
    }
    n->lo = lo;
    n->hi = hi;
    uint32_t count = end - begin;
    if (count <= ORC_LEAF_MAX) {
        n->left = begin;
        n->count = count;
        return;
    }
    /* binned SAH on the widest centroid axis */
    v3 ext = vsub(chi, clo);
    int ax = 0;
    if (ext.y > ext.x) ax = 1;
    if (ext.z > axis_of(ext, ax)) ax = 2;
    float amin = axis_of(clo, ax), aext = axis_of(ext, ax);
    uint32_t mid = begin + count / 2;
    if (aext > 0.0f) {
        v3 blo[ORC_BINS], bhi[ORC_BINS];
        uint32_t bcnt[ORC_BINS];
        for (int b = 0; b < ORC_BINS; ++b) {
            blo[b] = V(ORC_INF, ORC_INF, ORC_INF);
            bhi[b] = V(-ORC_INF, -ORC_INF, -ORC_INF);
            bcnt[b] = 0;
        }
        float scale = (float)ORC_BINS / aext;
        for (uint32_t i = begin; i < end; ++i) {
            uint32_t p = s->order[i];
            int b = (int)((axis_of(c->cen[p], ax) - amin) * scale);
            if (b < 0) b = 0;
            if (b >= ORC_BINS) b = ORC_BINS - 1;
            bcnt[b]++;
            blo[b] = V(fminf(blo[b].x, c->plo[p].x), fminf(blo[b].y, c->plo[p].y), fminf(blo[b].z, c->plo[p].z));
            bhi[b] = V(fmaxf(bhi[b].x, c->phi[p].x), fmaxf(bhi[b].y, c->phi[p].y), fmaxf(bhi[b].z, c->phi[p].z));
        }
        float right_area[ORC_BINS];
        uint32_t right_cnt[ORC_BINS];
        v3 rlo = V(ORC_INF, ORC_INF, ORC_INF), rhi = V(-ORC_INF, -ORC_INF, -ORC_INF);
        uint32_t rc = 0;
        for (int b = ORC_BINS - 1; b > 0; --b) {
            rlo = V(fminf(rlo.x, blo[b].x), fminf(rlo.y, blo[b].y), fminf(rlo.z, blo[b].z));
            rhi = V(fmaxf(rhi.x, bhi[b].x), fmaxf(rhi.y, bhi[b].y), fmaxf(rhi.z, bhi[b].z));
            rc += bcnt[b];
            right_area[b] = rc ? half_area(rlo, rhi) : 0.0f;
            right_cnt[b] = rc;
        }
        v3 llo = V(ORC_INF, ORC_INF, ORC_INF), lhi = V(-ORC_INF, -ORC_INF, -ORC_INF);
        uint32_t lc = 0;
        float best = ORC_INF;
        int best_b = -1;
        for (int b = 0; b < ORC_BINS - 1; ++b) {
            llo = V(fminf(llo.x, blo[b].x), fminf(llo.y, blo[b].y), fminf(llo.z, blo[b].z));
            lhi = V(fmaxf(lhi.x, bhi[b].x), fmaxf(lhi.y, bhi[b].y), fmaxf(lhi.z, bhi[b].z));
            lc += bcnt[b];
            if (lc == 0 || right_cnt[b + 1] == 0) continue;
            float cost = half_area(llo, lhi) * (float)lc + right_area[b + 1] * (float)right_cnt[b + 1];
            if (cost < best) {
                best = cost;
                best_b = b;
            }
        }
        if (best_b >= 0) {
            uint32_t i = begin, j = end;
            while (i < j) {
                uint32_t p = s->order[i];
                int b = (int)((axis_of(c->cen[p], ax) - amin) * scale);
                if (b < 0) b = 0;
                if (b >= ORC_BINS) b = ORC_BINS - 1;
                if (b <= best_b) {
                    ++i;
                } else {
                    --j;
                    s->order[i] = s->order[j];
                    s->order[j] = p;
                }
            }
            if (i > begin && i < end) mid = i;
        }
    }
    uint32_t left = (uint32_t)s->n_nodes;
    s->n_nodes += 2;
    n->left = left;
    n->count = 0;
    build_node(c, left, begin, mid);
    build_node(c, left + 1, mid, end);
}

int orc_scene_commit(orc_scene *s) {
    free(s->nodes);
    free(s->order);
    s->nodes = NULL;
    s->order = NULL;
    s->n_nodes = 0;
    s->lo = V(ORC_INF, ORC_INF, ORC_INF);
    s->hi = V(-ORC_INF, -ORC_INF, -ORC_INF);
    size_t n = s->n_prims;
    if (n > 0) {
        build_ctx c;
        c.s = s;
        c.plo = (v3 *)malloc(n * sizeof(v3));
        c.phi = (v3 *)malloc(n * sizeof(v3));
        c.cen = (v3 *)malloc(n * sizeof(v3));
        s->order = (uint32_t *)malloc(n * sizeof(uint32_t));
        s->nodes = (bvh_node *)malloc((2 * n) * sizeof(bvh_node));
        for (size_t i = 0; i < n; ++i) {
            v3 lo, hi;
            prim_bounds(&s->prims[i], &lo, &hi);
            s->lo = V(fminf(s->lo.x, lo.x), fminf(s->lo.y, lo.y), fminf(s->lo.z, lo.z));
            s->hi = V(fmaxf(s->hi.x, hi.x), fmaxf(s->hi.y, hi.y), fmaxf(s->hi.z, hi.z));
            c.cen[i] = V(0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z);
            pad_bounds(&lo, &hi);
            c.plo[i] = lo;
            c.phi[i] = hi;
            s->order[i] = (uint32_t)i;
        }
        s->n_nodes = 1;
        build_node(&c, 0, 0, (uint32_t)n);
        free(c.plo);
        free(c.phi);
        free(c.cen);
    }
    s->committed = 1;
    return 0;
}

/* embree.rs:146-161 */
void orc_scene_bounds(const orc_scene *s, float lo[3], float hi[3]) {
    v3_to(s->lo, lo);
    v3_to(s->hi, hi);
}

/* ------------------------------------------------------------------------------------------------ closest hit */

typedef struct {
    float t;
    uint32_t gidx; /* global primitive index, ORC_INVALID_ID = none */
    float u, v;
    v3 ng;
} closest_t;

/* Candidate order: smaller t wins; equal t goes to the smaller global primitive index, so the result does not
 * depend on the order primitives are visited in. */
static inline void test_prim(const orc_scene *s, uint32_t gi, v3 o, v3 d, float tnear, closest_t *best) {
    const prim_t *p = &s->prims[gi];
    float t, u = 0.0f, v = 0.0f;
    v3 ng;
    if (p->is_sphere) {
        if (!ray_sphere(o, d, tnear, best->t, p->v0, p->v1.x, &t, &ng)) return;
    } else {
        if (!ray_triangle(o, d, p->v0, p->v1, p->v2, &t, &u, &v)) return;
        if (!(t > tnear)) return;
        ng = V(0.0f, 0.0f, 0.0f);
    }
    if (t < best->t || (t == best->t && gi < best->gidx)) {
        best->t = t;
        best->gidx = gi;
        best->u = u;
        best->v = v;
        best->ng = ng;
    }
}

static void closest_brute(const orc_scene *s, v3 o, v3 d, float tnear, float tfar, closest_t *best) {
    best->t = tfar;
    best->gidx = ORC_INVALID_ID;
    for (size_t i = 0; i < s->n_prims; ++i) test_prim(s, (uint32_t)i, o, d, tnear, best);
}

static void closest_bvh(const orc_scene *s, v3 o, v3 d, float tnear, float tfar, closest_t *best) {
    best->t = tfar;
    best->gidx = ORC_INVALID_ID;
    if (s->n_nodes == 0) return;
    v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const bvh_node *n = &s->nodes[stack[--sp]];
        float t0, t1;
        if (!slab(n->lo, n->hi, o, inv, tnear, best->t, &t0, &t1)) continue;
        if (n->count) {
            for (uint32_t i = 0; i < n->count; ++i) test_prim(s, s->order[n->left + i], o, d, tnear, best);
        } else if (sp + 2 <= 128) {
            stack[sp++] = n->left;
            stack[sp++] = n->left + 1;
        }
    }
}

/* embree.rs:100-140 */
static int intersection_full(const orc_scene *s, v3 o, v3 d, float tnear, float tfar, int mode, orc_hit *h) {
    closest_t best;
    if (mode == ORC_ISECT_BVH && s->committed)
        closest_bvh(s, o, d, tnear, tfar, &best);
    else
        closest_brute(s, o, d, tnear, tfar, &best);
    if (best.gidx == ORC_INVALID_ID) {
        memset(h, 0, sizeof *h);
        h->t = ORC_INF;
        h->geom_id = h->prim_id = ORC_INVALID_ID;
        h->mat_id = 0;
        return 0;
    }
    const prim_t *p = &s->prims[best.gidx];
    v3 ng = p->is_sphere ? best.ng : vcross(vsub(p->v1, p->v0), vsub(p->v2, p->v0));
    v3 n = vnormalize_or_zero(ng);
    h->t = best.t;
    h->pos[0] = o.x + best.t * d.x;
    h->pos[1] = o.y + best.t * d.y;
    h->pos[2] = o.z + best.t * d.z;
    v3_to(n, h->n);
    h->uv[0] = best.u;
    h->uv[1] = best.v;
    h->geom_id = p->geom_id;
    h->prim_id = p->prim_id;
    h->mat_id = p->geom_id < s->n_geoms ? s->geom_material[p->geom_id] : 0;
    return 1;
}

void orc_intersect(const orc_scene *s, const orc_ray *rays, orc_hit *hits, size_t n, int mode) {
#pragma omp parallel for schedule(dynamic, 256)
    for (long long i = 0; i < (long long)n; ++i)
        intersection_full(s, v3_from(rays[i].org), v3_from(rays[i].dir), rays[i].tnear, rays[i].tfar, mode, &hits[i]);
}

/* ------------------------------------------------------------------------------------------------ camera */

/* glam Quat * Vec3 */
static v3 quat_mul_vec3(const float q[4], v3 v) {
    float w = q[3];
    v3 b = V(q[0], q[1], q[2]);
    float b2 = vdot(b, b);
    return vadd(vadd(vscale(w * w - b2, v), vscale(vdot(v, b) * 2.0f, b)), vscale(w * 2.0f, vcross(b, v)));
}

/* filter.rs:40-47 */
static float sample_tent(float c) { return c <= 0.5f ? sqrtf(2.0f * c) : 1.0f - sqrtf(2.0f - 2.0f * c); }

/* filter.rs:11-55: DummyFilter, BoxFilter, TriangleFilter, radius 0.7 on both axes (executor.rs:221-223) */
static float filter_offset(uint32_t kind, float p) {
    if (kind == ORC_FILTER_TRIANGLE) return 0.7f * (2.0f * sample_tent(p) - 1.0f);
    if (kind == ORC_FILTER_NONE) return p;
    return lerp_ext(p, -0.7f, 0.7f);
}

/* executor.rs:254-267 (pixel_worker up to camera.ray) + sampler.rs:18-112 + filter.rs:11-55 + camera.rs:62-85 */
static void camera_ray(const orc_camera *cam, const orc_render_params *p, uint32_t x, uint32_t y, uint32_t sample,
                       v3 *org, v3 *dir) {
    const uint32_t sampler = p->flags & 0x03u, filter = p->flags & 0x30u;
    uint32_t n = (uint32_t)floorf(sqrtf((float)p->spp)); /* executor.rs:218 */
    if (n == 0) n = 1;
    uint32_t index = sample % (n * n);
    float cx = (float)(index % n), cy = (float)(index / n);
    float r[4];
    orc_uniform4(p->seed, x, y, sample, 0, r);
    /* Sampler::sample_2d, twice: pixel position (executor.rs:256), then lens position (camera.rs:70) */
    float px, py, dx, dy;
    if (sampler == ORC_SAMPLER_UNIFORM) { /* sampler.rs:54-64 */
        px = r[0], py = r[1], dx = r[2], dy = r[3];
    } else if (sampler == ORC_SAMPLER_PIXEL_CENTRE) { /* sampler.rs:18-24 */
        px = py = dx = dy = 0.5f;
    } else { /* sampler.rs:91-102 */
        px = (cx + r[0]) / (float)n, py = (cy + r[1]) / (float)n;
        dx = (cx + r[2]) / (float)n, dy = (cy + r[3]) / (float)n;
    }
    float offx = filter_offset(filter, px), offy = filter_offset(filter, py);
    float cxp = ((float)x + 0.5f) + offx, cyp = ((float)y + 0.5f) + offy;
    float vx = 2.0f * (cxp / (float)cam->width) - 1.0f;
    float vy = 2.0f * (cyp / (float)cam->height) - 1.0f;
    v3 lens = v3_from(cam->lens);
    v3 sensor = V(lens.x + vx * cam->half_width, lens.y + vy * cam->half_height, lens.z + cam->focal_length);
    float ha = cam->aperture / 2.0f;
    v3 dst = V(lens.x + ha * dx, lens.y + ha * dy, lens.z + ha * 0.0f);
    *org = lens;
    *dir = vnormalize(quat_mul_vec3(cam->rot, vsub(dst, sensor)));
}

void orc_camera_ray(const orc_camera *cam, const orc_render_params *p, uint32_t x, uint32_t y, uint32_t sample,
                    orc_ray *ray) {
    v3 o, d;
    camera_ray(cam, p, x, y, sample, &o, &d);
    v3_to(o, ray->org);
    v3_to(d, ray->dir);
    ray->tnear = 0.0f;
    ray->tfar = ORC_INF;
}

/* ------------------------------------------------------------------------------------------------ integrators */

/* renderer.rs:14-22 */
typedef struct {
    v3 normal, position, albedo, color;
    float z, ray_depth;
    uint32_t samples;
} ray_result;

static const ray_result RAY_RESULT_DEFAULT = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, 0.0f, 0.0f, 0};

typedef struct {
    const orc_scene *s;
    const orc_render_params *p;
    uint32_t x, y, sample;
    uint32_t rays;
} path_ctx;

/* integrators/mod.rs:8-23 */
static ray_result sky_ray(void) {
    ray_result r = RAY_RESULT_DEFAULT;
    r.color = V(0.5f, 0.3f, 1.0f);
    r.samples = 1;
    return r;
}

/* pathtracing.rs:22-79, recursion kept as written */
static ray_result path_tracer_recursive(path_ctx *c, v3 org, v3 dir, float tfar, uint32_t depth) {
    if (depth == c->p->max_depth) return RAY_RESULT_DEFAULT;
    orc_hit h;
    c->rays++;
    if (!intersection_full(c->s, org, dir, 0.00001f, tfar, (int)c->p->isect_mode, &h)) return sky_ray();
    const orc_material *m = &c->s->materials[h.mat_id];
    v3 n = v3_from(h.n);
    float u[4];
    orc_uniform4(c->p->seed, c->x, c->y, c->sample, 1 + depth, u);
    bxdf_sample sm;
    if (!bsdf_sample_f(m, n, vneg(dir), u[0], u[1], u[2], &sm)) {
        sm.wi = V(0.0f, 0.0f, 0.0f);
        sm.f = V(0.0f, 0.0f, 0.0f);
        sm.pdf = 1.0f;
    }
    v3 fcos = vscale(fabsf(vdot(n, sm.wi)), sm.f);
    v3 le = material_le(m);
    v3 li;
    float ray_depth;
    if (fabsf(vmax_element(fcos)) != 0.0f) {
        ray_result rr = path_tracer_recursive(c, v3_from(h.pos), sm.wi, ORC_INF, depth + 1);
        li = vadd(le, vmul(vscale(1.0f / sm.pdf, fcos), rr.color));
        ray_depth = rr.ray_depth;
    } else {
        li = le;
        ray_depth = 0.0f;
    }
    ray_result out;
    out.normal = n;
    out.position = v3_from(h.pos);
    out.albedo = sm.f;
    out.color = li;
    out.z = h.t;
    out.ray_depth = ray_depth + h.t;
    out.samples = 1;
    return out;
}

/* The same estimator unrolled front to back, the form a wavefront renderer evaluates: throughput beta is carried
 * forward and the one radiance term a path can collect (sky or emitter; every other le() is zero) is weighted once.
 * Differs from the recursion only in the association of the f32 products and of the ray_depth sum. */
static ray_result path_tracer_iterative(path_ctx *c, v3 org, v3 dir) {
    ray_result out = RAY_RESULT_DEFAULT;
    uint32_t max_depth = c->p->max_depth;
    if (max_depth == 0) return out;
    v3 beta = V(1.0f, 1.0f, 1.0f);
    out.samples = 1;
    for (uint32_t depth = 0;; ++depth) {
        if (depth == max_depth) {
            out.color = vmul(beta, V(0.0f, 0.0f, 0.0f));
            break;
        }
        orc_hit h;
        c->rays++;
        if (!intersection_full(c->s, org, dir, 0.00001f, ORC_INF, (int)c->p->isect_mode, &h)) {
            out.color = vmul(beta, V(0.5f, 0.3f, 1.0f));
            break;
        }
        const orc_material *m = &c->s->materials[h.mat_id];
        v3 n = v3_from(h.n);
        float u[4];
        orc_uniform4(c->p->seed, c->x, c->y, c->sample, 1 + depth, u);
        bxdf_sample sm;
        if (!bsdf_sample_f(m, n, vneg(dir), u[0], u[1], u[2], &sm)) {
            sm.wi = V(0.0f, 0.0f, 0.0f);
            sm.f = V(0.0f, 0.0f, 0.0f);
            sm.pdf = 1.0f;
        }
        if (depth == 0) {
            out.normal = n;
            out.position = v3_from(h.pos);
            out.albedo = sm.f;
            out.z = h.t;
        }
        out.ray_depth = out.ray_depth + h.t;
        v3 fcos = vscale(fabsf(vdot(n, sm.wi)), sm.f);
        if (!(fabsf(vmax_element(fcos)) != 0.0f)) {
            out.color = vmul(beta, material_le(m));
            break;
        }
        beta = vmul(beta, vscale(1.0f / sm.pdf, fcos));
        org = v3_from(h.pos);
        dir = sm.wi;
    }
    return out;
}

/* randomwalk.rs:25-68 */
static ray_result random_walk(path_ctx *c, v3 org, v3 dir, float tfar, uint32_t depth) {
    if (depth == c->p->max_depth) return RAY_RESULT_DEFAULT;
    orc_hit h;
    c->rays++;
    if (!intersection_full(c->s, org, dir, 0.00001f, tfar, (int)c->p->isect_mode, &h)) return sky_ray();
    const orc_material *m = &c->s->materials[h.mat_id];
    v3 n = v3_from(h.n);
    float u[4];
    orc_uniform4(c->p->seed, c->x, c->y, c->sample, 1 + depth, u);
    v3 wi = uniform_unit_sphere3(u[0], u[1]);
    v3 f = bsdf_f(m, n, vneg(dir), wi);
    v3 fcos = vscale(fabsf(vdot(n, wi)), f);
    v3 le = material_le(m);
    v3 li;
    if (fabsf(vmax_element(fcos)) != 0.0f) {
        ray_result rr = random_walk(c, v3_from(h.pos), wi, ORC_INF, depth + 1);
        li = vadd(le, vmul(vscale(ORC_FRAC_1_PI / 4.0f, fcos), rr.color));
    } else {
        li = le;
    }
    ray_result out;
    out.normal = n;
    out.position = v3_from(h.pos);
    out.albedo = f;
    out.color = li;
    out.z = h.t;
    out.ray_depth = h.t;
    out.samples = 1;
    return out;
}

/* randomwalk.rs:25-68 unrolled front to back like path_tracer_iterative: no material of the reference both emits and
 * scatters (EmitBxDF::f is zero, every other le() is zero), so the one radiance term a walk collects is weighted by the
 * carried throughput once.  Differs from the recursion only in the association of the f32 products. */
static ray_result random_walk_iterative(path_ctx *c, v3 org, v3 dir) {
    ray_result out = RAY_RESULT_DEFAULT;
    uint32_t max_depth = c->p->max_depth;
    if (max_depth == 0) return out;
    v3 beta = V(1.0f, 1.0f, 1.0f);
    out.samples = 1;
    for (uint32_t depth = 0;; ++depth) {
        if (depth == max_depth) {
            out.color = vmul(beta, V(0.0f, 0.0f, 0.0f));
            break;
        }
        orc_hit h;
        c->rays++;
        if (!intersection_full(c->s, org, dir, 0.00001f, ORC_INF, (int)c->p->isect_mode, &h)) {
            out.color = vmul(beta, V(0.5f, 0.3f, 1.0f));
            break;
        }
        const orc_material *m = &c->s->materials[h.mat_id];
        v3 n = v3_from(h.n);
        float u[4];
        orc_uniform4(c->p->seed, c->x, c->y, c->sample, 1 + depth, u);
        v3 wi = uniform_unit_sphere3(u[0], u[1]);
        v3 f = bsdf_f(m, n, vneg(dir), wi);
        if (depth == 0) {
            out.normal = n;
            out.position = v3_from(h.pos);
            out.albedo = f;
            out.z = h.t;
            out.ray_depth = 0.0f + h.t; /* randomwalk.rs:65: the first hit's t only */
        }
        v3 fcos = vscale(fabsf(vdot(n, wi)), f);
        if (!(fabsf(vmax_element(fcos)) != 0.0f)) {
            out.color = vmul(beta, material_le(m));
            break;
        }
        beta = vmul(beta, vscale(ORC_FRAC_1_PI / 4.0f, fcos));
        org = v3_from(h.pos);
        dir = wi;
    }
    return out;
}

static ray_result render_one(const orc_scene *s, const orc_camera *cam, const orc_render_params *p, uint32_t x,
                             uint32_t y, uint32_t sample, uint32_t *rays) {
    path_ctx c;
    c.s = s;
    c.p = p;
    c.x = x;
    c.y = y;
    c.sample = sample;
    c.rays = 0;
    v3 org, dir;
    camera_ray(cam, p, x, y, sample, &org, &dir);
    ray_result r;
    if (p->integrator == ORC_INTEGRATOR_RANDOM_WALK)
        r = p->form == ORC_FORM_RECURSIVE ? random_walk(&c, org, dir, ORC_INF, 0) : random_walk_iterative(&c, org, dir);
    else if (p->form == ORC_FORM_RECURSIVE)
        r = path_tracer_recursive(&c, org, dir, ORC_INF, 0);
    else
        r = path_tracer_iterative(&c, org, dir);
    *rays = c.rays;
    return r;
}

uint32_t orc_render_sample(const orc_scene *s, const orc_camera *cam, const orc_render_params *p, uint32_t x,
                           uint32_t y, uint32_t sample, float out[15]) {
    uint32_t rays;
    ray_result r = render_one(s, cam, p, x, y, sample, &rays);
    v3_to(r.color, out);
    v3_to(r.normal, out + 3);
    v3_to(r.position, out + 6);
    v3_to(r.albedo, out + 9);
    out[12] = r.z;
    out[13] = r.ray_depth;
    out[14] = (float)r.samples;
    return rays;
}

/* stat.rs:92-95 */
static const float STUDENT_5[16] = {2.131f, 2.120f, 2.110f, 2.101f, 2.093f, 2.086f, 2.080f, 2.074f,
                                    2.069f, 2.064f, 2.060f, 2.056f, 2.052f, 2.048f, 2.045f, 2.042f};

static float variance_of(float count, float sum, float sqsum);

/* VarianceSeries::is_precise_enough, stat.rs:57-78 */
static int channel_precise(float count, float sum, float sqsum, float abs_err) {
    float df = count - 1.0f;
    if (df < 15.0f) return 0;
    int idx = (int)(df < 30.0f ? df : 30.0f) - 15;
    float err = STUDENT_5[idx] * sqrtf(variance_of(count, sum, sqsum));
    return err < abs_err;
}

/* executor.rs:207-252 (tile_worker's per-pixel sample loop) + renderer.rs:63-82 (add_sample, weight 1) */
uint64_t orc_render(const orc_scene *s, const orc_camera *cam, const orc_render_params *p, const uint32_t *xy,
                    size_t n_pixels, float *accum) {
    uint64_t total_rays = 0;
#ifdef _OPENMP
    int nthreads = p->threads ? (int)p->threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total_rays) num_threads(nthreads)
#endif
    for (long long i = 0; i < (long long)n_pixels; ++i) {
        uint32_t x = xy[2 * i], y = xy[2 * i + 1];
        float a[ORC_ACCUM_PLANES];
        for (int k = 0; k < ORC_ACCUM_PLANES; ++k) a[k] = accum[(size_t)k * n_pixels + (size_t)i];
        for (uint32_t sample = p->sample_begin; sample < p->sample_end; ++sample) {
            uint32_t rays;
            ray_result r = render_one(s, cam, p, x, y, sample, &rays);
            total_rays += rays;
            const float w = 1.0f; /* filter.rs:30 */
            const float col[3] = {r.color.x, r.color.y, r.color.z};
            for (int ch = 0; ch < 3; ++ch) {
                a[ch] += col[ch];                /* stat.rs:26 */
                a[3 + ch] += col[ch] * col[ch];  /* stat.rs:27 */
                a[6 + ch] = a[6 + ch] + w * col[ch]; /* stat.rs:167 */
            }
            a[9] += w; /* stat.rs:166 */
            a[10] += r.normal.x;
            a[11] += r.normal.y;
            a[12] += r.normal.z;
            a[13] = a[13] + r.position.x;
            a[14] = a[14] + r.position.y;
            a[15] = a[15] + r.position.z;
            a[16] = a[16] + r.albedo.x;
            a[17] = a[17] + r.albedo.y;
            a[18] = a[18] + r.albedo.z;
            a[19] += r.z;
            a[20] += r.ray_depth;
            a[21] += (float)r.samples;
            /* executor.rs:244-249: stop this pixel for the rest of the batch once all three channels are precise */
            if (p->allowed_error > 0.0f && channel_precise(a[21], a[0], a[3], p->allowed_error) &&
                channel_precise(a[21], a[1], a[4], p->allowed_error) &&
                channel_precise(a[21], a[2], a[5], p->allowed_error))
                break;
        }
        for (int k = 0; k < ORC_ACCUM_PLANES; ++k) accum[(size_t)k * n_pixels + (size_t)i] = a[k];
    }
    return total_rays;
}

/* ------------------------------------------------------------------------------------------------ resolve */

/* colorspace.rs:71-80 */
float orc_srgb_from_linear(float linear) {
    if (isnan(linear)) return 0.0f;
    linear = linear < 0.0f ? 0.0f : (linear > 1.0f ? 1.0f : linear);
    if (linear < 0.0031308f) return 12.92f * linear;
    return 1.055f * powf(linear, 1.0f / 2.4f) - 0.055f;
}

/* stat.rs:45-52 */
static float variance_of(float count, float sum, float sqsum) {
    if (count <= 2.0f) return ORC_INF;
    return (sqsum - sum * sum / count) / (count - 1.0f);
}

void orc_resolve(const float *accum, size_t n_pixels, float *out, int srgb) {
    for (size_t i = 0; i < n_pixels; ++i) {
#define A(k) accum[(size_t)(k) * n_pixels + i]
#define O(k) out[(size_t)(k) * n_pixels + i]
        float count = A(21);
        float inv = 1.0f / count; /* renderer.rs:49 */
        for (int c = 0; c < 3; ++c) {
            O(0 + c) = inv * A(10 + c);
            O(3 + c) = inv * A(13 + c);
            O(6 + c) = inv * A(16 + c);
            O(9 + c) = (A(9) == 0.0f) ? A(6 + c) : A(6 + c) / A(9); /* stat.rs:170-175 */
        }
        float vr = variance_of(count, A(0), A(3));
        float vg = variance_of(count, A(1), A(4));
        float vb = variance_of(count, A(2), A(5));
        O(12) = sqrtf(vr * vr + vg * vg + vb * vb); /* stat.rs:129-136 */
        O(13) = inv * A(19);
        O(14) = inv * A(20);
        if (srgb)
            for (int c = 0; c < 12; ++c) O(c) = orc_srgb_from_linear(O(c));
#undef A
#undef O
    }
}

/* ------------------------------------------------------------------------------------------------ tiles */

static uint32_t div_ceil_u32(uint32_t x, uint32_t y) { return x / y + (x % y != 0 ? 1u : 0u); }

uint32_t orc_tile_count(uint32_t width, uint32_t height, uint32_t grain) {
    return div_ceil_u32(width, grain) * div_ceil_u32(height, grain);
}

int orc_tile_get(uint32_t offset_x, uint32_t offset_y, uint32_t width, uint32_t height, uint32_t grain, uint32_t idx,
                 orc_tile *out) {
    if (idx >= orc_tile_count(width, height, grain)) return 0;
    uint32_t cols = div_ceil_u32(width, grain);
    uint32_t x = idx % cols, y = idx / cols;
    uint32_t xe = (x + 1) * grain, ye = (y + 1) * grain;
    out->x_start = offset_x + x * grain;
    out->x_end = offset_x + (width < xe ? width : xe);
    out->y_start = offset_y + y * grain;
    out->y_end = offset_y + (height < ye ? height : ye);
    return 1;
}
