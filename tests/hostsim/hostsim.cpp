// hostsim.cpp -- TEST HARNESS, not part of the product.  Compiles the per-ray device code (rt_device.h) and the
// host-side scene assembly (OBJ loader, BVH builder, flatten) with g++ so that they can be checked against the CPU
// oracle, and single-stepped under gdb, on a machine without a GPU.  The loops below stand in for the kernels' glue
// (queueing, compaction, accumulation order); the GPU tests cover the real kernels.
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#define RT_DEV inline
#include "../../raytracing_amd/csrc/bvh.h"
#include "../../raytracing_amd/csrc/obj_loader.h"
#include "../../raytracing_amd/csrc/rt_device.h"
#include "../../raytracing_amd/csrc/scene_build.h"

using namespace rtd;

namespace {

unsigned long long g_push_hist[64]; // pushes by stack position, for sizing the LDS part of the kernels' stack

struct VecStack {
    uint32_t data[256];
    int sp = 0;
    int high = 0;
    void push(uint32_t v) {
        ++g_push_hist[sp < 63 ? sp : 63];
        data[sp++] = v;
        if (sp > high) high = sp;
    }
    uint32_t pop() { return data[--sp]; }
    bool empty() const { return sp == 0; }
};

struct Scene {
    std::vector<Material> materials;
    std::vector<rts::HostPrim> prims;
    std::vector<uint32_t> geom_material;
    rtb::WideBvh bvh;
    float lo[3], hi[3];
    SceneView view{};
    int stack_high = 0;
};

struct Ray {
    float org[3], tnear, dir[3], tfar;
};
struct Hit {
    float t, pos[3], n[3], uv[2];
    uint32_t geom_id, prim_id, mat_id;
};
struct Params {
    uint32_t spp, sample_begin, sample_end, max_depth, integrator, samples_per_pass;
    uint64_t seed;
    float allowed_error;
    uint32_t flags;
};

} // namespace

extern "C" {

// pushes counted by stack position since the last call with reset != 0
void hs_push_hist(unsigned long long out[64], int reset) {
    for (int k = 0; k < 64; ++k) out[k] = g_push_hist[k];
    if (reset)
        for (int k = 0; k < 64; ++k) g_push_hist[k] = 0;
}

void *hs_scene_new() {
    Scene *s = new Scene();
    Material sky{};
    sky.kind = MAT_DIFFUSE;
    s->materials.push_back(sky);
    return s;
}
void hs_scene_free(void *p) { delete (Scene *)p; }

uint32_t hs_insert_material(void *p, uint32_t kind, const float rgb[3], float ior, float roughness) {
    Scene *s = (Scene *)p;
    Material m{};
    m.kind = kind;
    m.r = rgb[0];
    m.g = rgb[1];
    m.b = rgb[2];
    m.ior = ior;
    m.roughness = roughness;
    s->materials.push_back(m);
    return (uint32_t)s->materials.size() - 1;
}

uint32_t hs_insert_mesh(void *p, uint32_t material, const float *v, size_t nv, const uint32_t *idx, size_t nt) {
    Scene *s = (Scene *)p;
    (void)nv;
    uint32_t geom = (uint32_t)s->geom_material.size();
    s->geom_material.push_back(material);
    for (size_t i = 0; i < nt; ++i) {
        rts::HostPrim hp{};
        memcpy(hp.a, v + 3 * (size_t)idx[3 * i], 12);
        memcpy(hp.b, v + 3 * (size_t)idx[3 * i + 1], 12);
        memcpy(hp.c, v + 3 * (size_t)idx[3 * i + 2], 12);
        hp.geom = geom;
        hp.prim = (uint32_t)i;
        s->prims.push_back(hp);
    }
    return geom;
}

uint32_t hs_insert_sphere(void *p, uint32_t material, const float c[3], float r) {
    Scene *s = (Scene *)p;
    uint32_t geom = (uint32_t)s->geom_material.size();
    s->geom_material.push_back(material);
    rts::HostPrim hp{};
    memcpy(hp.a, c, 12);
    hp.b[0] = r;
    hp.geom = geom;
    hp.is_sphere = 1;
    s->prims.push_back(hp);
    return geom;
}

// Same steps as rtmi_scene_load_obj.  Returns the number of meshes inserted, -1 on error.
int hs_load_obj(void *p, const char *path, const float tr[3], const float sc[3], const float rot[4],
                uint32_t default_material) {
    Scene *s = (Scene *)p;
    rto::ObjFile obj;
    std::string err;
    if (!rto::load_obj(path, obj, err)) return -1;
    std::vector<uint32_t> ids;
    if (obj.materials_ok)
        for (auto &m : obj.materials) ids.push_back(hs_insert_material(p, MAT_DIFFUSE, m.diffuse, 0.0f, 0.0f));
    for (auto &model : obj.models) {
        uint32_t mat = default_material;
        if (obj.materials_ok && model.material_id >= 0 && (size_t)model.material_id < ids.size())
            mat = ids[(size_t)model.material_id];
        for (size_t v = 0; v + 2 < model.positions.size(); v += 3) rts::apply_transform(tr, sc, rot, &model.positions[v]);
        hs_insert_mesh(p, mat, model.positions.data(), model.positions.size() / 3, model.indices.data(),
                       model.indices.size() / 3);
    }
    (void)s;
    return (int)obj.models.size();
}

// out: n_prims, n_nodes, stack_need, depth
int hs_commit(void *p, uint32_t out[4]) {
    Scene *s = (Scene *)p;
    if (!rts::build_scene(s->prims, s->bvh, nullptr, s->lo, s->hi, nullptr)) return -1;
    s->view.nodes = s->bvh.items.data();
    s->view.prims = s->bvh.items.data();
    s->view.geom_material = s->geom_material.data();
    s->view.materials = s->materials.data();
    s->view.n_nodes = s->bvh.n_nodes;
    s->view.n_prims = (uint32_t)s->prims.size();
    s->view.n_geoms = (uint32_t)s->geom_material.size();
    s->view.n_materials = (uint32_t)s->materials.size();
    out[0] = s->view.n_prims;
    out[1] = s->view.n_nodes;
    out[2] = s->bvh.stack_need;
    out[3] = s->bvh.depth;
    return 0;
}

// Leaf records of the committed tree (a record cut by spatial splits counts once per part) and the number of splits.
void hs_tree_stats(void *p, uint32_t out[2]) {
    Scene *s = (Scene *)p;
    out[0] = (uint32_t)s->bvh.prim_order.size();
    out[1] = s->bvh.spatial_splits;
}

void hs_bounds(void *p, float lo[3], float hi[3]) {
    Scene *s = (Scene *)p;
    memcpy(lo, s->lo, 12);
    memcpy(hi, s->hi, 12);
}

int hs_stack_high(void *p) { return ((Scene *)p)->stack_high; }

// geometry handles and per-mesh data, for loader parity checks: returns triangle count of geometry g and copies up to
// max_tris triangles (9 floats each, v0 v1 v2)
size_t hs_geometry_triangles(void *p, uint32_t g, float *out, size_t max_tris) {
    Scene *s = (Scene *)p;
    size_t n = 0;
    for (auto &hp : s->prims) {
        if (hp.geom != g || hp.is_sphere) continue;
        if (n < max_tris) {
            memcpy(out + 9 * n, hp.a, 12);
            memcpy(out + 9 * n + 3, hp.b, 12);
            memcpy(out + 9 * n + 6, hp.c, 12);
        }
        ++n;
    }
    return n;
}
uint32_t hs_geometry_material(void *p, uint32_t g) { return ((Scene *)p)->geom_material[g]; }
uint32_t hs_geometry_count(void *p) { return (uint32_t)((Scene *)p)->geom_material.size(); }
uint32_t hs_material_count(void *p) { return (uint32_t)((Scene *)p)->materials.size(); }
void hs_material_get(void *p, uint32_t i, uint32_t *kind, float out[5]) {
    const Material &m = ((Scene *)p)->materials[i];
    *kind = m.kind;
    out[0] = m.r;
    out[1] = m.g;
    out[2] = m.b;
    out[3] = m.ior;
    out[4] = m.roughness;
}

void hs_intersect(void *p, const Ray *rays, Hit *hits, size_t n) {
    Scene *s = (Scene *)p;
    for (size_t i = 0; i < n; ++i) {
        V3 o = mk(rays[i].org[0], rays[i].org[1], rays[i].org[2]);
        V3 d = mk(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
        VecStack st;
        Closest best;
        closest_hit(s->view, o, d, rays[i].tnear, rays[i].tfar, st, best, (uint32_t *)nullptr, (uint32_t *)nullptr);
        if (st.high > s->stack_high) s->stack_high = st.high;
        HitInfo h;
        finish_hit(s->view, o, d, best, h);
        Hit &out = hits[i];
        out.t = h.t;
        out.pos[0] = h.pos.x;
        out.pos[1] = h.pos.y;
        out.pos[2] = h.pos.z;
        out.n[0] = h.n.x;
        out.n[1] = h.n.y;
        out.n[2] = h.n.z;
        out.uv[0] = h.u;
        out.uv[1] = h.v;
        out.geom_id = h.geom_id;
        out.prim_id = h.prim_id;
        out.mat_id = h.mat_id;
    }
}

// One pixel sample through the same per-vertex steps the extend, shade and tail kernels run.  out as in the oracle's
// orc_render_sample: color 0-2, normal 3-5, position 6-8, albedo 9-11, z, ray_depth, samples.  Returns rays traced.
static uint32_t render_sample(Scene *s, const CameraView &cam, const Params &prm, uint32_t x, uint32_t y,
                              uint32_t sample, float out[15]) {
    memset(out, 0, 15 * sizeof(float));
    if (prm.max_depth == 0) return 0;
    uint32_t strata = (uint32_t)floorf(sqrtf((float)prm.spp));
    if (strata == 0) strata = 1;
    V3 org, dir, beta = mk(1.0f, 1.0f, 1.0f);
    camera_ray(cam, strata, prm.flags, prm.seed, x, y, sample, org, dir);
    uint32_t rays = 0;
    float ray_depth = 0.0f;
    V3 radiance = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t depth = 0; depth < prm.max_depth; ++depth) {
        VecStack st;
        Closest best;
        closest_hit(s->view, org, dir, RT_TNEAR, INFINITY, st, best, (uint32_t *)nullptr, (uint32_t *)nullptr);
        if (st.high > s->stack_high) s->stack_high = st.high;
        ++rays;
        HitInfo h;
        finish_hit(s->view, org, dir, best, h);
        VertexOut vo;
        bool cont = shade_vertex(s->view, h, prm.integrator, prm.seed, x, y, sample, depth, org, dir, beta, vo);
        if (depth == 0) {
            out[3] = vo.normal.x;
            out[4] = vo.normal.y;
            out[5] = vo.normal.z;
            out[6] = vo.position.x;
            out[7] = vo.position.y;
            out[8] = vo.position.z;
            out[9] = vo.albedo.x;
            out[10] = vo.albedo.y;
            out[11] = vo.albedo.z;
            out[12] = vo.z;
            ray_depth = 0.0f + vo.t_add;
        } else if (h.hit && vo.t_add != 0.0f) {
            ray_depth = ray_depth + vo.t_add;
        }
        if (cont && depth + 1 == prm.max_depth) {
            vo.radiance = mul(beta, mk(0.0f, 0.0f, 0.0f));
            cont = false;
        }
        if (!cont) {
            radiance = vo.radiance;
            break;
        }
    }
    out[0] = radiance.x;
    out[1] = radiance.y;
    out[2] = radiance.z;
    out[13] = ray_depth;
    out[14] = 1.0f;
    return rays;
}

uint32_t hs_render_sample(void *p, const CameraView *cam, const Params *prm, uint32_t x, uint32_t y, uint32_t sample,
                          float out[15]) {
    return render_sample((Scene *)p, *cam, *prm, x, y, sample, out);
}

uint64_t hs_render(void *p, const CameraView *cam, const Params *prm, const uint32_t *xy, size_t npix, float *accum) {
    Scene *s = (Scene *)p;
    uint64_t rays = 0;
    for (size_t i = 0; i < npix; ++i) {
        float a[22];
        for (int k = 0; k < 22; ++k) a[k] = accum[(size_t)k * npix + i];
        for (uint32_t sample = prm->sample_begin; sample < prm->sample_end; ++sample) {
            float r[15];
            rays += render_sample(s, *cam, *prm, xy[2 * i], xy[2 * i + 1], sample, r);
            if (prm->max_depth == 0) continue;
            for (int ch = 0; ch < 3; ++ch) {
                a[ch] += r[ch];
                a[3 + ch] += r[ch] * r[ch];
                a[6 + ch] = a[6 + ch] + 1.0f * r[ch];
            }
            a[9] += 1.0f;
            for (int k = 0; k < 9; ++k) a[10 + k] = a[10 + k] + r[3 + k];
            a[19] += r[12];
            a[20] += r[13];
            a[21] += 1.0f;
        }
        for (int k = 0; k < 22; ++k) accum[(size_t)k * npix + i] = a[k];
    }
    return rays;
}

// Per-bounce work statistics over a pixel list: out[3 * d + 0..2] = rays, node visits, primitive tests at bounce d.
void hs_depth_stats(void *p, const CameraView *cam, const Params *prm, const uint32_t *xy, size_t npix, uint32_t max_d,
                    double *out) {
    Scene *s = (Scene *)p;
    uint32_t strata = (uint32_t)floorf(sqrtf((float)prm->spp));
    if (strata == 0) strata = 1;
    for (size_t i = 0; i < npix; ++i)
        for (uint32_t sample = prm->sample_begin; sample < prm->sample_end; ++sample) {
            V3 org, dir, beta = mk(1.0f, 1.0f, 1.0f);
            camera_ray(*cam, strata, prm->flags, prm->seed, xy[2 * i], xy[2 * i + 1], sample, org, dir);
            for (uint32_t depth = 0; depth < prm->max_depth; ++depth) {
                VecStack st;
                Closest best;
                uint32_t nodes = 0, tests = 0;
                closest_hit(s->view, org, dir, RT_TNEAR, INFINITY, st, best, &nodes, &tests);
                if (depth < max_d) {
                    out[3 * depth] += 1;
                    out[3 * depth + 1] += nodes;
                    out[3 * depth + 2] += tests;
                }
                HitInfo h;
                finish_hit(s->view, org, dir, best, h);
                VertexOut vo;
                if (!shade_vertex(s->view, h, prm->integrator, prm->seed, xy[2 * i], xy[2 * i + 1], sample, depth, org,
                                  dir, beta, vo))
                    break;
            }
        }
}

const void *hs_view(void *p) { return &((Scene *)p)->view; }

} // extern "C"

// ---------------------------------------------------------------------------------------------- wave scheduling model
//
// Replays the traversal kernel's lane scheduling for waves of 64 rays on the host, with an issue-cost model, to compare
// scheduling policies before writing them for the GPU.  policy 0: every step the wave runs the phase (node / record)
// most busy lanes wait for.  policy 1: a lane that reaches a record parks it in a queue of `depth` entries and goes on
// with the next stack entry; records are tested when `leaf_lanes` lanes have parked records, a lane is blocked, or no
// lane has a node to visit.  Finished lanes are refilled when fewer than refill_below lanes are busy.
// out: issue cost, useful cost, node steps, record steps, node phases, record phases, refills
extern "C" void hs_wave_sim(void *p, const Ray *rays, size_t n, int policy, int refill_below, int depth, int leaf_lanes,
                            const double cost[4], double out[8]) {
    Scene *s = (Scene *)p;
    const double c_node = cost[0], c_leaf = cost[1], c_refill = cost[2], c_iter = cost[3];
    struct Lane {
        bool has = false;
        V3 o, d;
        Trav tr;
        VecStack st;
        uint32_t pend[8];
        int np = 0;
    };
    for (int k = 0; k < 8; ++k) out[k] = 0.0;
    size_t next = 0;
    while (next < n) {
        // one persistent wave takes up to 4096 rays (a claimed range), then the next wave starts
        const size_t wave_end = std::min(n, next + 4096);
        std::vector<Lane> L(64);
        for (;;) {
            int n_busy = 0;
            for (auto &l : L)
                if (l.has && (l.tr.cur != RT_EMPTY || l.np > 0)) ++n_busy;
            if (n_busy < refill_below || n_busy == 0) {
                bool any = false;
                for (auto &l : L) {
                    if (l.has && l.tr.cur == RT_EMPTY && l.np == 0) l.has = false, any = true;
                    if (!l.has && next < wave_end) {
                        const Ray &r = rays[next++];
                        l.o = mk(r.org[0], r.org[1], r.org[2]);
                        l.d = mk(r.dir[0], r.dir[1], r.dir[2]);
                        l.st = VecStack();
                        trav_begin(s->view, l.o, l.d, INFINITY, l.tr);
                        l.np = 0;
                        l.has = true;
                        any = true;
                    }
                }
                if (any) out[0] += 64.0 * c_refill, out[6] += 1.0;
                n_busy = 0;
                for (auto &l : L)
                    if (l.has && (l.tr.cur != RT_EMPTY || l.np > 0)) ++n_busy;
                if (n_busy == 0) {
                    if (next >= wave_end) break;
                    continue;
                }
            }
            out[0] += 64.0 * c_iter;
            if (policy >= 1) {
                // park records (bounded work per iteration: two moves)
                for (auto &l : L)
                    for (int m = 0; m < 2; ++m)
                        if (l.has && l.tr.cur != RT_EMPTY && (l.tr.cur & RT_LEAF) && l.np < depth) {
                            l.pend[l.np++] = l.tr.cur;
                            l.tr.cur = l.st.empty() ? RT_EMPTY : l.st.pop();
                        }
            }
            int n_node = 0, n_leaf = 0, n_blocked = 0;
            for (auto &l : L) {
                if (!l.has) continue;
                const bool at_node = l.tr.cur != RT_EMPTY && !(l.tr.cur & RT_LEAF);
                if (at_node) ++n_node;
                if (policy == 0) {
                    if (l.tr.cur != RT_EMPTY && (l.tr.cur & RT_LEAF)) ++n_leaf;
                } else {
                    if (l.np > 0) ++n_leaf;
                    if (l.tr.cur != RT_EMPTY && (l.tr.cur & RT_LEAF)) ++n_blocked; // queue full
                }
            }
            bool node_phase;
            if (policy == 0)
                node_phase = n_leaf == 0 || (n_node > 0 && n_node >= n_leaf);
            else if (policy == 1)
                node_phase = n_node > 0 && n_blocked == 0 && n_leaf < leaf_lanes;
            else
                node_phase = n_node > 0 && n_leaf < leaf_lanes; // blocked lanes wait
            if (node_phase) {
                out[0] += 64.0 * c_node;
                out[4] += 1.0;
                for (auto &l : L)
                    if (l.has && l.tr.cur != RT_EMPTY && !(l.tr.cur & RT_LEAF)) {
                        trav_node_step(s->view, RT_TNEAR, l.tr, l.st);
                        out[1] += c_node;
                        out[2] += 1.0;
                    }
            } else {
                out[0] += 64.0 * c_leaf;
                out[5] += 1.0;
                for (auto &l : L) {
                    if (!l.has) continue;
                    if (policy == 0) {
                        if (l.tr.cur != RT_EMPTY && (l.tr.cur & RT_LEAF)) {
                            trav_leaf_step(s->view, l.o, l.d, RT_TNEAR, l.tr, l.st);
                            out[1] += c_leaf;
                            out[3] += 1.0;
                        }
                    } else if (l.np > 0) {
                        // oldest parked record first (it was the nearest when parked)
                        const uint32_t ref = l.pend[0];
                        for (int k = 1; k < l.np; ++k) l.pend[k - 1] = l.pend[k];
                        --l.np;
                        test_prim(s->view, ref & 0x0FFFFFFFu, l.o, l.d, RT_TNEAR, l.tr.best);
                        out[1] += c_leaf;
                        out[3] += 1.0;
                    }
                }
            }
        }
    }
}


// Variant of the model above with two rays per lane: every step a lane advances whichever of its two rays waits for the
// phase the wave runs.  out as hs_wave_sim.  cost[3] is charged per iteration and should include the selects that pick
// the active ray's registers.
extern "C" void hs_wave_sim2(void *p, const Ray *rays, size_t n, int refill_below, const double cost[4], double out[8]) {
    Scene *s = (Scene *)p;
    const double c_node = cost[0], c_leaf = cost[1], c_refill = cost[2], c_iter = cost[3];
    struct Slot {
        bool has = false;
        V3 o, d;
        Trav tr;
        VecStack st;
    };
    for (int k = 0; k < 8; ++k) out[k] = 0.0;
    size_t next = 0;
    while (next < n) {
        const size_t wave_end = std::min(n, next + 8192);
        std::vector<Slot> L(128);
        for (;;) {
            int n_busy = 0;
            for (auto &l : L)
                if (l.has && l.tr.cur != RT_EMPTY) ++n_busy;
            if (n_busy < refill_below || n_busy == 0) {
                bool any = false;
                for (auto &l : L) {
                    if (l.has && l.tr.cur == RT_EMPTY) l.has = false, any = true;
                    if (!l.has && next < wave_end) {
                        const Ray &r = rays[next++];
                        l.o = mk(r.org[0], r.org[1], r.org[2]);
                        l.d = mk(r.dir[0], r.dir[1], r.dir[2]);
                        l.st = VecStack();
                        trav_begin(s->view, l.o, l.d, INFINITY, l.tr);
                        l.has = true;
                        any = true;
                    }
                }
                if (any) out[0] += 64.0 * c_refill, out[6] += 1.0;
                n_busy = 0;
                for (auto &l : L)
                    if (l.has && l.tr.cur != RT_EMPTY) ++n_busy;
                if (n_busy == 0) {
                    if (next >= wave_end) break;
                    continue;
                }
            }
            out[0] += 64.0 * c_iter;
            int lanes_node = 0, lanes_leaf = 0;
            auto at_node = [](const Slot &l) { return l.has && l.tr.cur != RT_EMPTY && !(l.tr.cur & RT_LEAF); };
            auto at_leaf = [](const Slot &l) { return l.has && l.tr.cur != RT_EMPTY && (l.tr.cur & RT_LEAF); };
            for (int l = 0; l < 64; ++l) {
                if (at_node(L[l]) || at_node(L[l + 64])) ++lanes_node;
                if (at_leaf(L[l]) || at_leaf(L[l + 64])) ++lanes_leaf;
            }
            const bool node_phase = lanes_leaf == 0 || (lanes_node > 0 && lanes_node >= lanes_leaf);
            if (node_phase) {
                out[0] += 64.0 * c_node;
                out[4] += 1.0;
                for (int l = 0; l < 64; ++l) {
                    Slot &x = at_node(L[l]) ? L[l] : L[l + 64];
                    if (at_node(x)) {
                        trav_node_step(s->view, RT_TNEAR, x.tr, x.st);
                        out[1] += c_node;
                        out[2] += 1.0;
                    }
                }
            } else {
                out[0] += 64.0 * c_leaf;
                out[5] += 1.0;
                for (int l = 0; l < 64; ++l) {
                    Slot &x = at_leaf(L[l]) ? L[l] : L[l + 64];
                    if (at_leaf(x)) {
                        trav_leaf_step(s->view, x.o, x.d, RT_TNEAR, x.tr, x.st);
                        out[1] += c_leaf;
                        out[3] += 1.0;
                    }
                }
            }
        }
    }
}

// Model of a block of four waves that share 256 ray slots kept in LDS: slot s belongs to lane position s % 64, and
// every iteration a wave runs one phase (inner-node step or record test) in which each of its lanes works on a slot of
// its lane position that waits for that phase and that no other wave holds.  A lane keeps its slot while the slot's ray
// stays in the phase the wave runs; otherwise it puts the slot back and takes another (a swap: state written to and
// read from LDS).  Waves are stepped round-robin.  cost = {node step, record step, refill pass, per iteration, swap};
// out[0] wave-time (sum over waves), out[1] useful lane-time, out[2] node lane-steps, out[3] record lane-steps,
// out[4] node iterations, out[5] record iterations, out[6] refill passes, out[7] iterations with a swap.
extern "C" void hs_block_sim(void *p, const Ray *rays, size_t n, int refill_below, int n_waves, const double cost[5],
                             double out[8]) {
    Scene *s = (Scene *)p;
    const double c_node = cost[0], c_leaf = cost[1], c_refill = cost[2], c_iter = cost[3], c_swap = cost[4];
    struct Slot {
        bool has = false;
        int owner = -1; // wave holding it
        V3 o, d;
        Trav tr;
        VecStack st;
    };
    for (int k = 0; k < 8; ++k) out[k] = 0.0;
    const int W = n_waves;
    size_t next = 0;
    while (next < n) {
        const size_t block_end = std::min(n, next + (size_t)4096 * W);
        std::vector<Slot> S((size_t)64 * W);
        std::vector<std::vector<int>> cur(W, std::vector<int>(64, -1)); // slot each wave-lane holds
        auto phase_of = [](const Slot &x) { // 0 none/finished, 1 node, 2 leaf
            if (!x.has || x.tr.cur == RT_EMPTY) return 0;
            return (x.tr.cur & RT_LEAF) ? 2 : 1;
        };
        bool all_done = false;
        while (!all_done) {
            all_done = true;
            for (int w = 0; w < W; ++w) {
                // lanes of this wave that could work in each phase (their own slot or a free one at their position)
                int can_node = 0, can_leaf = 0, live = 0, refillable = 0;
                for (int l = 0; l < 64; ++l) {
                    bool n_ok = false, l_ok = false, r_ok = false;
                    for (int k = 0; k < W; ++k) {
                        const Slot &x = S[(size_t)l + 64 * k];
                        if (x.owner != -1 && x.owner != w) continue;
                        const int ph = phase_of(x);
                        n_ok |= ph == 1;
                        l_ok |= ph == 2;
                        r_ok |= ph == 0 && (x.has || next < block_end);
                    }
                    can_node += n_ok;
                    can_leaf += l_ok;
                    live += n_ok || l_ok;
                    refillable += r_ok;
                }
                if (refillable > 64 - (refill_below < 0 ? 48 : refill_below) || live == 0) {
                    // flush + refill the finished slots nobody holds (or that this wave holds) at every lane position
                    bool any = false;
                    for (int l = 0; l < 64; ++l)
                        for (int k = 0; k < W; ++k) {
                            Slot &x = S[(size_t)l + 64 * k];
                            if (x.owner != -1 && x.owner != w) continue;
                            if (x.has && x.tr.cur == RT_EMPTY) {
                                x.has = false;
                                any = true;
                                if (x.owner == w) cur[w][l] = -1, x.owner = -1;
                            }
                            if (!x.has && next < block_end) {
                                const Ray &r = rays[next++];
                                x.o = mk(r.org[0], r.org[1], r.org[2]);
                                x.d = mk(r.dir[0], r.dir[1], r.dir[2]);
                                x.st = VecStack();
                                trav_begin(s->view, x.o, x.d, INFINITY, x.tr);
                                x.has = true;
                                any = true;
                            }
                        }
                    if (any) {
                        out[0] += 64.0 * c_refill, out[6] += 1.0;
                        all_done = false;
                        continue; // next wave; this one decides its phase next round
                    }
                    if (live == 0) continue; // nothing left for this wave
                }
                all_done = false;
                bool node_phase = can_leaf == 0 || (can_node > 0 && can_node >= can_leaf);
                if (refill_below < 0) { // specialised waves: the last wave tests records whenever enough lanes can
                    const int thr = -refill_below;
                    if (w == W - 1) node_phase = can_leaf < thr && can_node > 0;
                    else node_phase = can_node > 0 ? (can_leaf < 56 || can_node >= can_leaf) : false;
                    if (can_leaf == 0) node_phase = true;
                }
                const int want = node_phase ? 1 : 2;
                bool swapped = false;
                int active = 0;
                for (int l = 0; l < 64; ++l) {
                    int c = cur[w][l];
                    if (c >= 0 && phase_of(S[(size_t)c]) != want) {
                        S[(size_t)c].owner = -1; // put back
                        cur[w][l] = c = -1;
                        swapped = true;
                    }
                    if (c < 0)
                        for (int k = 0; k < W; ++k) {
                            Slot &x = S[(size_t)l + 64 * k];
                            if (x.owner == -1 && phase_of(x) == want) {
                                x.owner = w;
                                cur[w][l] = c = l + 64 * k;
                                swapped = true;
                                break;
                            }
                        }
                    if (c < 0) continue;
                    Slot &x = S[(size_t)c];
                    ++active;
                    if (node_phase)
                        trav_node_step(s->view, RT_TNEAR, x.tr, x.st);
                    else
                        trav_leaf_step(s->view, x.o, x.d, RT_TNEAR, x.tr, x.st);
                }
                const double c_step = node_phase ? c_node : c_leaf;
                out[0] += 64.0 * (c_step + c_iter + (swapped ? c_swap : 0.0));
                out[1] += active * c_step;
                out[node_phase ? 2 : 3] += active;
                out[node_phase ? 4 : 5] += 1.0;
                out[7] += swapped ? 1.0 : 0.0;
            }
        }
    }
}

// Model of a wave that defers its record tests into a queue shared by its 64 lanes: a lane that reaches a leaf appends
// (lane, record) to the queue and goes on with its next stack entry; a record phase takes up to 64 queue entries, one
// per lane whatever lane the entry came from (ray and best candidate are read from / merged into the owner's state, as
// they would be through LDS), so a record phase runs with every lane busy whenever 64 entries are waiting.  A lane
// with no node left is flushed only once all its entries have been tested.  Candidates are merged by (t, primitive
// index), so the deferred order does not change the result; what changes is that inner-node steps see a best distance
// that lags behind (less culling), which the model reproduces because it runs the real traversal.
// cost = {node step, record step, refill pass, per iteration, post pass}; fire = queue length that triggers a record
// phase; cap = queue capacity; a record phase also runs when fewer than `low` lanes have a node and the queue holds
// more entries than that.  out as hs_wave_sim; out[7] = record phases with fewer than 64 entries.
extern "C" void hs_queue_sim(void *p, const Ray *rays, size_t n, int refill_below, int fire, int cap, int low,
                             const double cost[5], double out[8]) {
    Scene *s = (Scene *)p;
    const double c_node = cost[0], c_leaf = cost[1], c_refill = cost[2], c_iter = cost[3], c_post = cost[4];
    struct Lane {
        bool has = false;
        V3 o, d;
        Trav tr;
        VecStack st;
        uint64_t last_post = 0; // sequence number after this lane's latest entry
    };
    struct Entry {
        int lane;
        uint32_t slot;
    };
    for (int k = 0; k < 8; ++k) out[k] = 0.0;
    size_t next = 0;
    while (next < n) {
        const size_t wave_end = std::min(n, next + 4096);
        std::vector<Lane> L(64);
        std::vector<Entry> q;
        uint64_t q_done = 0, q_posted = 0;
        size_t q_head = 0;
        auto pending = [&](const Lane &l) { return l.last_post > q_done; };
        for (;;) {
            int n_live = 0;
            for (auto &l : L)
                if (l.has && (l.tr.cur != RT_EMPTY || pending(l))) ++n_live;
            if (n_live < refill_below || n_live == 0) {
                bool any = false;
                for (auto &l : L) {
                    if (l.has && l.tr.cur == RT_EMPTY && !pending(l)) l.has = false, any = true;
                    if (!l.has && next < wave_end) {
                        const Ray &r = rays[next++];
                        l.o = mk(r.org[0], r.org[1], r.org[2]);
                        l.d = mk(r.dir[0], r.dir[1], r.dir[2]);
                        l.st = VecStack();
                        trav_begin(s->view, l.o, l.d, INFINITY, l.tr);
                        l.last_post = 0;
                        l.has = true;
                        any = true;
                    }
                }
                if (any) out[0] += 64.0 * c_refill, out[6] += 1.0;
                n_live = 0;
                for (auto &l : L)
                    if (l.has && (l.tr.cur != RT_EMPTY || pending(l))) ++n_live;
                if (n_live == 0) {
                    if (next >= wave_end) break;
                    continue;
                }
            }
            out[0] += 64.0 * c_iter;
            // post pass: every lane at a leaf appends one record (while the queue has room) and pops
            bool posted = false;
            for (int i = 0; i < 64; ++i) {
                Lane &l = L[i];
                if (!(l.has && l.tr.cur != RT_EMPTY && (l.tr.cur & RT_LEAF))) continue;
                if ((int)(q.size() - q_head) >= cap) break;
                const uint32_t first = l.tr.cur & 0x0FFFFFFFu, more = (l.tr.cur >> 28) & 7u;
                q.push_back({i, first});
                l.last_post = ++q_posted;
                posted = true;
                if (more != 0u)
                    l.tr.cur = RT_LEAF | ((more - 1u) << 28) | (first + 1u);
                else
                    l.tr.cur = l.st.empty() ? RT_EMPTY : l.st.pop();
            }
            if (posted) out[0] += 64.0 * c_post;
            int n_node = 0, n_leaf = 0;
            for (auto &l : L) {
                if (!l.has || l.tr.cur == RT_EMPTY) continue;
                if (l.tr.cur & RT_LEAF)
                    ++n_leaf; // posts in the next iteration
                else
                    ++n_node;
            }
            const int q_len = (int)(q.size() - q_head);
            const bool full = q_len + n_leaf > cap;
            const bool record_phase = q_len >= fire || full || (n_node == 0 && n_leaf == 0 && q_len > 0) || (n_node < q_len && n_node < low);
            if (!record_phase) {
                if (n_node == 0) continue; // lanes at leaves post in the next iteration
                out[0] += 64.0 * c_node;
                out[4] += 1.0;
                for (auto &l : L)
                    if (l.has && l.tr.cur != RT_EMPTY && !(l.tr.cur & RT_LEAF)) {
                        trav_node_step(s->view, RT_TNEAR, l.tr, l.st);
                        out[1] += c_node;
                        out[2] += 1.0;
                    }
            } else {
                const int take = std::min(q_len, 64);
                out[0] += 64.0 * c_leaf;
                out[5] += 1.0;
                if (take < 64) out[7] += 1.0;
                for (int k = 0; k < take; ++k) {
                    const Entry e = q[q_head++];
                    Lane &l = L[e.lane];
                    test_prim(s->view, e.slot, l.o, l.d, RT_TNEAR, l.tr.best);
                    out[1] += c_leaf;
                    out[3] += 1.0;
                }
                q_done += (uint64_t)take;
                if (q_head == q.size()) q.clear(), q_head = 0;
            }
        }
    }
}

// Every item (inner node: its index; primitive record: RT_LEAF | slot) a ray's closest-hit traversal fetches, in visit
// order, for layout studies: ray i's items are items[offsets[i] .. offsets[i + 1]).  Returns the number of items the
// rays fetch in total; items beyond `cap` are counted but not stored.
extern "C" size_t hs_trace(void *p, const Ray *rays, size_t n, uint32_t *items, size_t cap, uint64_t *offsets) {
    Scene *s = (Scene *)p;
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        offsets[i] = total;
        V3 o = mk(rays[i].org[0], rays[i].org[1], rays[i].org[2]);
        V3 d = mk(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
        VecStack st;
        Trav tr;
        trav_begin(s->view, o, d, rays[i].tfar, tr);
        while (tr.cur != RT_EMPTY) {
            if (!(tr.cur & RT_LEAF)) {
                if (total < cap) items[total] = tr.cur;
                ++total;
                trav_node_step(s->view, rays[i].tnear, tr, st);
            } else {
                if (total < cap) items[total] = RT_LEAF | (tr.cur & 0x0FFFFFFFu);
                ++total;
                trav_leaf_step(s->view, o, d, rays[i].tnear, tr, st);
            }
        }
    }
    offsets[n] = total;
    return total;
}

// The item array of the tree (nodes and primitive records, 16 words each); returns the number of items.  Item 0 is
// the root; the children of a node are neighbours (rt_device.h describes the node words).
extern "C" size_t hs_items(void *p, uint32_t *out, size_t max_items) {
    Scene *s = (Scene *)p;
    const size_t n = s->bvh.n_items < max_items ? s->bvh.n_items : max_items;
    if (out) memcpy(out, s->bvh.items.data(), n * 64);
    return s->bvh.n_items;
}

// The rays real paths trace at bounce `depth_wanted` (the path tracer's own bounce rays, not synthetic ones): runs
// every sample of the listed pixels and stores the ray entering that bounce.  Returns the number of rays stored.
extern "C" size_t hs_path_rays(void *p, const CameraView *cam, const Params *prm, const uint32_t *xy, size_t npix,
                               uint32_t depth_wanted, Ray *out, size_t cap) {
    Scene *s = (Scene *)p;
    uint32_t strata = (uint32_t)floorf(sqrtf((float)prm->spp));
    if (strata == 0) strata = 1;
    size_t n = 0;
    for (size_t i = 0; i < npix && n < cap; ++i)
        for (uint32_t sample = prm->sample_begin; sample < prm->sample_end && n < cap; ++sample) {
            V3 org, dir, beta = mk(1.0f, 1.0f, 1.0f);
            camera_ray(*cam, strata, prm->flags, prm->seed, xy[2 * i], xy[2 * i + 1], sample, org, dir);
            for (uint32_t depth = 0; depth < prm->max_depth; ++depth) {
                if (depth == depth_wanted) {
                    Ray &r = out[n++];
                    r.org[0] = org.x, r.org[1] = org.y, r.org[2] = org.z;
                    r.dir[0] = dir.x, r.dir[1] = dir.y, r.dir[2] = dir.z;
                    r.tnear = RT_TNEAR;
                    r.tfar = INFINITY;
                    break;
                }
                VecStack st;
                Closest best;
                closest_hit(s->view, org, dir, RT_TNEAR, INFINITY, st, best, (uint32_t *)nullptr, (uint32_t *)nullptr);
                HitInfo h;
                finish_hit(s->view, org, dir, best, h);
                VertexOut vo;
                if (!shade_vertex(s->view, h, prm->integrator, prm->seed, xy[2 * i], xy[2 * i + 1], sample, depth, org,
                                  dir, beta, vo))
                    break;
            }
        }
    return n;
}
