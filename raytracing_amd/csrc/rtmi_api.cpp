// rtmi_api.cpp -- the C ABI of include/rtmi.h: scene assembly, OBJ loading, commit (BVH build + upload), batch
// intersection and the wavefront render driver.  Host side only; the kernels are in kernels.hip.
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "bvh.h"
#include "obj_loader.h"
#include "rtmi_internal.h"

using rts::HostPrim;
using rti::DeviceScope;
using rti::fail;

namespace {

thread_local std::string g_error;

// Device-time stopwatch of one call: intervals are bracketed by pairs of HIP events on one stream, and collect(),
// called after the host has synchronised with that stream, adds every finished interval to its total and takes the
// events back for reuse.  However many intervals a call times, it holds only as many events as are in flight
// between two synchronisation points.
struct Stopwatch {
    struct Interval {
        hipEvent_t begin, end;
        float *total;
        bool closed;
        int id;
    };
    std::vector<hipEvent_t> idle;
    std::vector<Interval> pending;
    int next_id = 0;
    bool failed = false;
    hipEvent_t take() {
        if (!idle.empty()) {
            hipEvent_t e = idle.back();
            idle.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) {
            failed = true;
            return nullptr;
        }
        return e;
    }
    // Starts an interval on `stream` whose length is added to *total; returns its id for stop(), -1 on failure.
    int start(hipStream_t stream, float *total) {
        Interval iv{take(), take(), total, false, next_id};
        if (!iv.begin || !iv.end) {
            if (iv.begin) idle.push_back(iv.begin);
            if (iv.end) idle.push_back(iv.end);
            return -1;
        }
        (void)hipEventRecord(iv.begin, stream);
        pending.push_back(iv);
        return next_id++;
    }
    void stop(int id, hipStream_t stream) {
        for (Interval &iv : pending)
            if (iv.id == id && !iv.closed) {
                (void)hipEventRecord(iv.end, stream);
                iv.closed = true;
            }
    }
    void stop_all(hipStream_t stream) {
        for (Interval &iv : pending)
            if (!iv.closed) {
                (void)hipEventRecord(iv.end, stream);
                iv.closed = true;
            }
    }
    // Adds up and recycles the closed intervals whose end has been reached on the device.
    void collect() {
        std::vector<Interval> open;
        for (const Interval &iv : pending) {
            if (!iv.closed || hipEventQuery(iv.end) != hipSuccess) {
                open.push_back(iv);
                continue;
            }
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, iv.begin, iv.end) == hipSuccess) *iv.total += ms;
            idle.push_back(iv.begin);
            idle.push_back(iv.end);
        }
        pending.swap(open);
    }
    ~Stopwatch() {
        for (const Interval &iv : pending) {
            (void)hipEventDestroy(iv.begin);
            (void)hipEventDestroy(iv.end);
        }
        for (hipEvent_t e : idle) (void)hipEventDestroy(e);
    }
};

} // namespace

int rti::fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

namespace {

void free_scene_device(rtmi_scene *s) {
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_prims) (void)hipFree(s->d_prims);
    if (s->d_geom_material) (void)hipFree(s->d_geom_material);
    if (s->d_materials) (void)hipFree(s->d_materials);
    if (s->d_spill) (void)hipFree(s->d_spill);
    s->d_spill = nullptr;
    for (auto &slot : s->slots) {
        if (slot.d) (void)hipFree(slot.d);
        if (slot.done) (void)hipEventDestroy(slot.done);
        slot = rtmi_scene::SpillSlot{};
    }
    if (s->d_stage_rays) (void)hipFree(s->d_stage_rays);
    if (s->d_stage_hits) (void)hipFree(s->d_stage_hits);
    if (s->h_stage_rays) (void)hipHostFree(s->h_stage_rays);
    if (s->h_stage_hits) (void)hipHostFree(s->h_stage_hits);
    s->d_stage_rays = s->h_stage_rays = nullptr;
    s->d_stage_hits = s->h_stage_hits = nullptr;
    s->stage_cap = s->pinned_cap = 0;
    s->d_nodes = s->d_prims = nullptr;
    s->d_geom_material = nullptr;
    s->d_materials = nullptr;
}

} // namespace

extern "C" {

const char *rtmi_last_error(void) { return g_error.c_str(); }
const char *rtmi_version(void) { return "rtmi 0.1 (gfx950)"; }

int rtmi_device_create(int hip_ordinal, rtmi_device **out) {
    if (!out) return fail(RTMI_ERR_ARG, "rtmi_device_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available");
    int ordinal = hip_ordinal;
    if (ordinal < 0) HIP_TRY(hipGetDevice(&ordinal));
    if (ordinal >= count) return fail(RTMI_ERR_ARG, "HIP device ordinal out of range");
    HIP_TRY(hipSetDevice(ordinal));
    rtmi_device *d = new rtmi_device();
    d->ordinal = ordinal;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) == hipSuccess && prop.multiProcessorCount > 0)
        d->n_cu = prop.multiProcessorCount;
    e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete d;
        return fail(RTMI_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    e = hipMalloc((void **)&d->d_visits, 16);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(d->stream);
        delete d;
        return fail(RTMI_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    (void)hipMemset(d->d_visits, 0, 16);
    if (int se = rtk::static_spill_address(&d->static_spill, &d->static_spill_words)) {
        (void)hipFree(d->d_visits);
        (void)hipStreamDestroy(d->stream);
        delete d;
        return fail(RTMI_ERR_DEVICE, "hipGetSymbolAddress(static stack-overflow buffer) failed with code " +
                                         std::to_string(se));
    }
    *out = d;
    return RTMI_OK;
}

void rtmi_device_destroy(rtmi_device *dev) {
    if (!dev) return;
    rtmi_device_release(dev);
}

int rtmi_device_count_visits(rtmi_device *dev, int enable) {
    if (!dev) return fail(RTMI_ERR_ARG, "device is NULL");
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    dev->count_visits = enable != 0;
    HIP_TRY(hipMemsetAsync(dev->d_visits, 0, 16, dev->stream));
    HIP_TRY(hipStreamSynchronize(dev->stream));
    return RTMI_OK;
}

int rtmi_device_get_visits(rtmi_device *dev, uint64_t out[2]) {
    if (!dev || !out) return fail(RTMI_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    HIP_TRY(hipStreamSynchronize(dev->stream));
    HIP_TRY(hipMemcpy(out, dev->d_visits, 16, hipMemcpyDeviceToHost));
    return RTMI_OK;
}

int rtmi_device_last_render_profile(rtmi_device *dev, rtmi_bounce_profile *out, uint32_t capacity, uint32_t *n_out) {
    if (!dev || !n_out || (capacity && !out)) return fail(RTMI_ERR_ARG, "rtmi_device_last_render_profile: NULL argument");
    std::lock_guard<std::mutex> g(dev->lock);
    *n_out = (uint32_t)dev->profile.size();
    for (uint32_t i = 0; i < capacity && i < dev->profile.size(); ++i) out[i] = dev->profile[i];
    return RTMI_OK;
}

void *rtmi_device_stream(rtmi_device *dev) { return dev ? (void *)dev->stream : nullptr; }

int rtmi_scene_create(rtmi_device *dev, rtmi_scene **out) {
    if (!dev || !out) return fail(RTMI_ERR_ARG, "rtmi_scene_create: NULL argument");
    rtmi_scene *s = new rtmi_scene();
    s->dev = dev;
    dev->refs.fetch_add(1);
    rtd::Material sky{}; // embree.rs:41-46
    sky.kind = rtd::MAT_DIFFUSE;
    s->materials.push_back(sky);
    *out = s;
    return RTMI_OK;
}

void rtmi_scene_destroy(rtmi_scene *scene) {
    if (!scene) return;
    rtmi_device *dev = scene->dev;
    {
        DeviceScope scope(dev->ordinal);
        (void)hipStreamSynchronize(dev->stream);
        free_scene_device(scene);
    }
    delete scene;
    rtmi_device_release(dev);
}

int rtmi_scene_insert_material(rtmi_scene *scene, const rtmi_material *mat, uint32_t *out_id) {
    if (!scene || !mat) return fail(RTMI_ERR_ARG, "rtmi_scene_insert_material: NULL argument");
    if (scene->committed) return fail(RTMI_ERR_STATE, "scene already committed");
    if (mat->kind > RTMI_MAT_EMIT) return fail(RTMI_ERR_ARG, "unknown material kind");
    rtd::Material m{};
    m.kind = mat->kind;
    m.r = mat->rgb[0];
    m.g = mat->rgb[1];
    m.b = mat->rgb[2];
    m.ior = mat->ior;
    m.roughness = mat->roughness;
    if (out_id) *out_id = (uint32_t)scene->materials.size();
    scene->materials.push_back(m);
    return RTMI_OK;
}

int rtmi_scene_insert_light(rtmi_scene *scene, const float pos[3]) {
    if (!scene || !pos) return fail(RTMI_ERR_ARG, "rtmi_scene_insert_light: NULL argument");
    if (scene->committed) return fail(RTMI_ERR_STATE, "scene already committed");
    scene->lights.insert(scene->lights.end(), pos, pos + 3);
    return RTMI_OK;
}

int rtmi_scene_insert_mesh(rtmi_scene *scene, uint32_t material, const float *vertices, size_t n_vertices,
                           const uint32_t *indices, size_t n_triangles, uint32_t *out_geom) {
    if (!scene || (n_vertices && !vertices) || (n_triangles && !indices))
        return fail(RTMI_ERR_ARG, "rtmi_scene_insert_mesh: NULL argument");
    if (scene->committed) return fail(RTMI_ERR_STATE, "scene already committed");
    if (material >= scene->materials.size()) return fail(RTMI_ERR_ARG, "unknown material id");
    if (scene->prims.size() + n_triangles >= (1u << 28)) return fail(RTMI_ERR_LIMIT, "more than 2^28 primitives");
    if (scene->geom_material.size() >= rtd::RT_GEOM_MASK) return fail(RTMI_ERR_LIMIT, "more than 2^24 geometries");
    for (size_t i = 0; i < 3 * n_triangles; ++i)
        if (indices[i] >= n_vertices) return fail(RTMI_ERR_ARG, "vertex index out of range");
    const uint32_t geom = (uint32_t)scene->geom_material.size();
    scene->geom_material.push_back(material);
    scene->prims.reserve(scene->prims.size() + n_triangles);
    for (size_t i = 0; i < n_triangles; ++i) {
        HostPrim p{};
        memcpy(p.a, vertices + 3 * (size_t)indices[3 * i], 12);
        memcpy(p.b, vertices + 3 * (size_t)indices[3 * i + 1], 12);
        memcpy(p.c, vertices + 3 * (size_t)indices[3 * i + 2], 12);
        p.geom = geom;
        p.prim = (uint32_t)i;
        p.is_sphere = 0;
        scene->prims.push_back(p);
    }
    if (out_geom) *out_geom = geom;
    return RTMI_OK;
}

int rtmi_scene_insert_sphere(rtmi_scene *scene, uint32_t material, const float center[3], float radius,
                             uint32_t *out_geom) {
    if (!scene || !center) return fail(RTMI_ERR_ARG, "rtmi_scene_insert_sphere: NULL argument");
    if (scene->committed) return fail(RTMI_ERR_STATE, "scene already committed");
    if (material >= scene->materials.size()) return fail(RTMI_ERR_ARG, "unknown material id");
    if (scene->prims.size() + 1 >= (1u << 28)) return fail(RTMI_ERR_LIMIT, "more than 2^28 primitives");
    if (scene->geom_material.size() >= rtd::RT_GEOM_MASK) return fail(RTMI_ERR_LIMIT, "more than 2^24 geometries");
    const uint32_t geom = (uint32_t)scene->geom_material.size();
    scene->geom_material.push_back(material);
    HostPrim p{};
    memcpy(p.a, center, 12);
    p.b[0] = radius;
    p.geom = geom;
    p.prim = 0;
    p.is_sphere = 1;
    scene->prims.push_back(p);
    if (out_geom) *out_geom = geom;
    return RTMI_OK;
}

int rtmi_scene_load_obj(rtmi_scene *scene, const char *path, const rtmi_transform *transform,
                        uint32_t default_material, uint32_t *out_first_geom, uint32_t *out_n_geoms) {
    if (!scene || !path || !transform) return fail(RTMI_ERR_ARG, "rtmi_scene_load_obj: NULL argument");
    if (scene->committed) return fail(RTMI_ERR_STATE, "scene already committed");
    if (default_material >= scene->materials.size()) return fail(RTMI_ERR_ARG, "unknown default material id");
    rto::ObjFile obj;
    std::string err;
    if (!rto::load_obj(path, obj, err)) return fail(RTMI_ERR_IO, err);
    // loader/obj.rs:36-91: one Diffuse(Kd) per MTL material, in file order
    std::vector<uint32_t> ids;
    if (obj.materials_ok) {
        for (const auto &m : obj.materials) {
            rtmi_material d{};
            d.kind = RTMI_MAT_DIFFUSE;
            memcpy(d.rgb, m.diffuse, 12);
            uint32_t id = 0;
            if (int rc = rtmi_scene_insert_material(scene, &d, &id)) return rc;
            ids.push_back(id);
        }
    }
    const uint32_t first = (uint32_t)scene->geom_material.size();
    for (auto &model : obj.models) {
        uint32_t mat = default_material; // loader/obj.rs:100-107
        if (obj.materials_ok && model.material_id >= 0 && (size_t)model.material_id < ids.size())
            mat = ids[(size_t)model.material_id];
        for (size_t v = 0; v + 2 < model.positions.size(); v += 3) rts::apply_transform(transform->translation, transform->scale, transform->rot, &model.positions[v]);
        if (int rc = rtmi_scene_insert_mesh(scene, mat, model.positions.data(), model.positions.size() / 3,
                                            model.indices.data(), model.indices.size() / 3, nullptr))
            return rc;
    }
    if (out_first_geom) *out_first_geom = first;
    if (out_n_geoms) *out_n_geoms = (uint32_t)scene->geom_material.size() - first;
    return RTMI_OK;
}

int rtmi_scene_commit(rtmi_scene *scene, rtmi_progress_fn progress, void *user) {
    if (!scene) return fail(RTMI_ERR_ARG, "rtmi_scene_commit: scene is NULL");
    if (scene->committed) return fail(RTMI_ERR_STATE, "scene already committed");
    rtmi_device *dev = scene->dev;
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t n = scene->prims.size();
    rtb::WideBvh bvh;
    size_t n_records = 0;
    std::function<bool(double)> cb;
    if (progress) cb = [progress, user](double a) { return progress(a, user) != 0; };
    if (!rts::build_scene(scene->prims, bvh, &n_records, scene->lo, scene->hi, cb))
        return fail(RTMI_ERR_CANCELLED, "commit cancelled by the progress callback");
    // the quantised child boxes span at most 255 steps of 2^40; a non-finite coordinate lands here too (a NaN corner is
    // left out of the bounds, and no ray hits its triangle anyway)
    for (int k = 0; k < 3 && n > 0; ++k)
        if (!(fabsf(scene->lo[k]) <= 1e14f) || !(fabsf(scene->hi[k]) <= 1e14f))
            return fail(RTMI_ERR_LIMIT, "scene bounds beyond 1e14 units (or not finite): not representable in the tree's boxes");
    if (bvh.n_items >= rtd::RT_MAX_ITEMS)
        return fail(RTMI_ERR_LIMIT, "scene too large: more than 2^26 BVH nodes and primitive records");
    if (bvh.stack_need > rtk::MAX_STACK)
        return fail(RTMI_ERR_LIMIT, "BVH needs a deeper traversal stack than the kernels provide");
    free_scene_device(scene);
    // nodes and primitive records live in one array of 64-byte items (see bvh.h)
    const size_t item_bytes = bvh.items.size() * sizeof(rtd::F4);
    const size_t node_bytes = (size_t)bvh.n_nodes * 64;
    const size_t prim_bytes = item_bytes - node_bytes; // records and the padding of the layout
    if (item_bytes) {
        HIP_TRY(hipMalloc((void **)&scene->d_nodes, item_bytes));
        HIP_TRY(hipMemcpy(scene->d_nodes, bvh.items.data(), item_bytes, hipMemcpyHostToDevice));
    }
    scene->d_prims = nullptr; // records are addressed through the same base as nodes
    const size_t gm_bytes = std::max<size_t>(1, scene->geom_material.size()) * sizeof(uint32_t);
    HIP_TRY(hipMalloc((void **)&scene->d_geom_material, gm_bytes));
    if (!scene->geom_material.empty())
        HIP_TRY(hipMemcpy(scene->d_geom_material, scene->geom_material.data(),
                          scene->geom_material.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&scene->d_materials, scene->materials.size() * sizeof(rtd::Material)));
    HIP_TRY(hipMemcpy(scene->d_materials, scene->materials.data(), scene->materials.size() * sizeof(rtd::Material),
                      hipMemcpyHostToDevice));
    // resident grid of the traversal kernel (its LDS stack and registers bound the blocks per CU)
    {
        const uint32_t per_cu = (uint32_t)rtk::extend_blocks_per_cu(false), per_cu_q = (uint32_t)rtk::extend_blocks_per_cu(true);
        scene->grid_blocks = (uint32_t)dev->n_cu * std::max(per_cu, per_cu_q);
        scene->grid_blocks_vote = (uint32_t)dev->n_cu * per_cu;
        scene->grid_blocks_queue = (uint32_t)dev->n_cu * per_cu_q;
        scene->grid_blocks_tail = std::min(scene->grid_blocks, (uint32_t)dev->n_cu * (uint32_t)rtk::tail_blocks_per_cu());
        const size_t deep = bvh.stack_need > (uint32_t)rtk::SPILL_FROM ? bvh.stack_need - rtk::SPILL_FROM : 0;
        scene->spill_bytes = std::max<size_t>(16, deep * scene->grid_blocks * rtk::BLOCK * sizeof(uint32_t));
        HIP_TRY(hipMalloc((void **)&scene->d_spill, scene->spill_bytes));
    }
    scene->view.nodes = scene->d_nodes;
    scene->view.prims = scene->d_nodes;
    scene->view.geom_material = scene->d_geom_material;
    scene->view.materials = scene->d_materials;
    scene->view.n_nodes = bvh.n_nodes;
    scene->view.n_prims = (uint32_t)n;
    scene->view.n_geoms = (uint32_t)scene->geom_material.size();
    scene->view.n_materials = (uint32_t)scene->materials.size();
    scene->info.n_prims = n;
    scene->info.n_nodes = bvh.n_nodes;
    scene->info.node_bytes = node_bytes;
    scene->info.prim_bytes = prim_bytes;
    scene->info.n_geoms = scene->view.n_geoms;
    scene->info.n_materials = scene->view.n_materials;
    scene->info.stack_need = bvh.stack_need;
    scene->info.max_depth = bvh.depth;
    scene->info.sah_cost = bvh.sah_cost;
    scene->info.n_records = (uint32_t)n_records;
    scene->info.spatial_splits = bvh.spatial_splits;
    scene->info.build_ms =
        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    scene->committed = true;
    if (progress) progress(1.0, user);
    return RTMI_OK;
}

int rtmi_scene_bounds(const rtmi_scene *scene, float lo[3], float hi[3]) {
    if (!scene || !lo || !hi) return fail(RTMI_ERR_ARG, "rtmi_scene_bounds: NULL argument");
    if (!scene->committed) return fail(RTMI_ERR_STATE, "scene not committed");
    memcpy(lo, scene->lo, 12);
    memcpy(hi, scene->hi, 12);
    return RTMI_OK;
}

int rtmi_scene_get_info(const rtmi_scene *scene, rtmi_scene_info *out) {
    if (!scene || !out) return fail(RTMI_ERR_ARG, "rtmi_scene_get_info: NULL argument");
    if (!scene->committed) return fail(RTMI_ERR_STATE, "scene not committed");
    *out = scene->info;
    return RTMI_OK;
}

int rtmi_scene_geometry_material(const rtmi_scene *scene, uint32_t geom, uint32_t *out_material) {
    if (!scene || !out_material) return fail(RTMI_ERR_ARG, "rtmi_scene_geometry_material: NULL argument");
    if (geom >= scene->geom_material.size()) return fail(RTMI_ERR_ARG, "unknown geometry id");
    *out_material = scene->geom_material[geom];
    return RTMI_OK;
}

int rtmi_intersect_device(rtmi_scene *scene, const rtmi_ray *d_rays, rtmi_hit *d_hits, size_t n, void *hip_stream) {
    if (!scene || (n && (!d_rays || !d_hits))) return fail(RTMI_ERR_ARG, "rtmi_intersect_device: NULL argument");
    if (!scene->committed) return fail(RTMI_ERR_STATE, "scene not committed");
    if (n == 0) return RTMI_OK;
    USE_DEVICE(scene->dev);
    static_assert(sizeof(rtmi_ray) == sizeof(rtk::RayAoS) && sizeof(rtmi_hit) == sizeof(rtk::HitAoS), "layout");
    hipStream_t stream = (hipStream_t)hip_stream;
    if (scene->info.stack_need <= (uint32_t)rtk::LDS_STACK) {
        // the whole stack fits LDS: launches share nothing
        if (rtk::launch_intersect(scene->view, reinterpret_cast<const rtk::RayAoS *>(d_rays),
                                  reinterpret_cast<rtk::HitAoS *>(d_hits), n, scene->d_spill, scene->grid_blocks, stream))
            return fail(RTMI_ERR_DEVICE, "intersect kernel launch failed");
        return RTMI_OK;
    }
    std::lock_guard<std::mutex> g(scene->query_lock);
    rtmi_scene::SpillSlot &slot = scene->slots[scene->next_slot % rtmi_scene::SPILL_SLOTS];
    if (!slot.d) {
        HIP_TRY(hipMalloc((void **)&slot.d, scene->spill_bytes));
        hipError_t e = hipEventCreateWithFlags(&slot.done, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipFree(slot.d);
            slot.d = nullptr;
            return fail(RTMI_ERR_DEVICE, std::string("hipEventCreate: ") + hipGetErrorString(e));
        }
    }
    if (slot.used) HIP_TRY(hipStreamWaitEvent(stream, slot.done, 0));
    if (rtk::launch_intersect(scene->view, reinterpret_cast<const rtk::RayAoS *>(d_rays),
                              reinterpret_cast<rtk::HitAoS *>(d_hits), n, slot.d, scene->grid_blocks, stream))
        return fail(RTMI_ERR_DEVICE, "intersect kernel launch failed");
    HIP_TRY(hipEventRecord(slot.done, stream));
    slot.used = true;
    ++scene->next_slot;
    return RTMI_OK;
}

int rtmi_intersect_device_static(rtmi_scene *scene, const rtmi_ray *d_rays, rtmi_hit *d_hits, size_t n,
                                 void *hip_stream) {
    if (!scene || (n && (!d_rays || !d_hits)))
        return fail(RTMI_ERR_ARG, "rtmi_intersect_device_static: NULL argument");
    if (!scene->committed) return fail(RTMI_ERR_STATE, "scene not committed");
    if (n == 0) return RTMI_OK;
    rtmi_device *dev = scene->dev;
    USE_DEVICE(dev);
    // the grid is cut down to what the fixed buffer covers: `deep` overflow entries per launched thread
    uint32_t grid_cap = scene->grid_blocks;
    if (scene->info.stack_need > RTMI_STATIC_MAX_STACK)
        return fail(RTMI_ERR_LIMIT, "rtmi_intersect_device_static: the tree needs " +
                                        std::to_string(scene->info.stack_need) + " traversal stack entries, more than "
                                        "RTMI_STATIC_MAX_STACK; use rtmi_intersect_device");
    if (scene->info.stack_need > (uint32_t)rtk::LDS_STACK) {
        const size_t deep = scene->info.stack_need - rtk::LDS_STACK;
        const size_t fit = dev->static_spill_words / (deep * rtk::BLOCK);
        if (fit == 0) return fail(RTMI_ERR_LIMIT, "rtmi_intersect_device_static: the static buffer is too small");
        grid_cap = (uint32_t)std::min<size_t>(grid_cap, fit);
    }
    if (rtk::launch_intersect(scene->view, reinterpret_cast<const rtk::RayAoS *>(d_rays),
                              reinterpret_cast<rtk::HitAoS *>(d_hits), n, dev->static_spill, grid_cap,
                              (hipStream_t)hip_stream))
        return fail(RTMI_ERR_DEVICE, "intersect kernel launch failed");
    return RTMI_OK;
}

// Largest |component| a ray origin may have for the culling boxes to stay conservative: boxes are padded by 1e-6 of
// the scene's largest |coordinate| (scene_build.cpp), which covers the rounding of the slab distances for origins
// within 16 times that.
static float origin_limit(const rtmi_scene *scene) {
    float m = 0.0f;
    for (int k = 0; k < 3; ++k) m = fmaxf(m, fmaxf(fabsf(scene->lo[k]), fabsf(scene->hi[k])));
    return 16.0f * m;
}

int rtmi_intersect(rtmi_scene *scene, const rtmi_ray *rays, rtmi_hit *hits, size_t n) {
    if (!scene || (n && (!rays || !hits))) return fail(RTMI_ERR_ARG, "rtmi_intersect: NULL argument");
    if (!scene->committed) return fail(RTMI_ERR_STATE, "scene not committed");
    if (n == 0) return RTMI_OK;
    if (!scene->prims.empty()) {
        const float limit = origin_limit(scene);
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k)
                if (fabsf(rays[i].org[k]) > limit) // NaN passes: such a ray misses everything anyway
                    return fail(RTMI_ERR_ARG, "rtmi_intersect: ray " + std::to_string(i) + " starts more than 16 scene "
                                "extents from the origin; move its origin along the ray towards the scene");
    }
    rtmi_device *dev = scene->dev;
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    // device staging pair, kept between calls
    if (scene->stage_cap < n) {
        if (scene->d_stage_rays) (void)hipFree(scene->d_stage_rays);
        if (scene->d_stage_hits) (void)hipFree(scene->d_stage_hits);
        scene->d_stage_rays = nullptr;
        scene->d_stage_hits = nullptr;
        scene->stage_cap = 0;
        const size_t cap = std::max<size_t>(n, 1024);
        HIP_TRY(hipMalloc((void **)&scene->d_stage_rays, cap * sizeof(rtmi_ray)));
        HIP_TRY(hipMalloc((void **)&scene->d_stage_hits, cap * sizeof(rtmi_hit)));
        scene->stage_cap = cap;
    }
    // small batches (the per-ray use of Shape::intersection_full) go through pinned host memory: one memcpy each way
    // instead of the runtime's staging of pageable memory
    constexpr size_t PINNED_RAYS = 1024;
    const bool pinned = n <= PINNED_RAYS;
    if (pinned && scene->pinned_cap == 0) {
        HIP_TRY(hipHostMalloc((void **)&scene->h_stage_rays, PINNED_RAYS * sizeof(rtmi_ray), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void **)&scene->h_stage_hits, PINNED_RAYS * sizeof(rtmi_hit), hipHostMallocDefault));
        scene->pinned_cap = PINNED_RAYS;
    }
    const rtmi_ray *src = rays;
    rtmi_hit *dst = hits;
    if (pinned) {
        memcpy(scene->h_stage_rays, rays, n * sizeof(rtmi_ray));
        src = scene->h_stage_rays;
        dst = scene->h_stage_hits;
    }
    HIP_TRY(hipMemcpyAsync(scene->d_stage_rays, src, n * sizeof(rtmi_ray), hipMemcpyHostToDevice, dev->stream));
    // on the library's stream, under the device lock: the render's spill buffer is free
    if (rtk::launch_intersect(scene->view, reinterpret_cast<const rtk::RayAoS *>(scene->d_stage_rays),
                              reinterpret_cast<rtk::HitAoS *>(scene->d_stage_hits), n, scene->d_spill, scene->grid_blocks,
                              dev->stream)) {
        (void)hipStreamSynchronize(dev->stream);
        return fail(RTMI_ERR_DEVICE, "intersect kernel launch failed");
    }
    hipError_t e = hipMemcpyAsync(dst, scene->d_stage_hits, n * sizeof(rtmi_hit), hipMemcpyDeviceToHost, dev->stream);
    hipError_t e2 = hipStreamSynchronize(dev->stream);
    if (e != hipSuccess) return fail(RTMI_ERR_DEVICE, std::string("copy: ") + hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(RTMI_ERR_DEVICE, std::string("intersect: ") + hipGetErrorString(e2));
    if (pinned) memcpy(hits, scene->h_stage_hits, n * sizeof(rtmi_hit));
    return RTMI_OK;
}

int rtmi_film_create(rtmi_device *dev, const uint32_t *xy, size_t n_pixels, float *d_accum, rtmi_film **out) {
    if (!dev || !out || (n_pixels && !xy)) return fail(RTMI_ERR_ARG, "rtmi_film_create: NULL argument");
    if (n_pixels >= (1ull << 30)) return fail(RTMI_ERR_LIMIT, "2^30 pixels or more in one film");
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    rtmi_film *f = new rtmi_film();
    f->dev = dev;
    static std::atomic<unsigned long long> next_film_id{1};
    f->id = next_film_id.fetch_add(1);
    f->n_pixels = n_pixels;
    const size_t xy_bytes = std::max<size_t>(1, n_pixels) * 2 * sizeof(uint32_t);
    hipError_t e = hipMalloc((void **)&f->d_xy, xy_bytes);
    if (e == hipSuccess && n_pixels) e = hipMemcpy(f->d_xy, xy, n_pixels * 2 * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (d_accum) {
            f->d_accum = d_accum;
        } else {
            const size_t bytes = std::max<size_t>(1, n_pixels) * RTMI_ACCUM_PLANES * sizeof(float);
            e = hipMalloc((void **)&f->d_accum, bytes);
            if (e == hipSuccess) {
                f->own_accum = true;
                e = hipMemset(f->d_accum, 0, bytes);
            }
        }
    }
    if (e != hipSuccess) {
        if (f->d_xy) (void)hipFree(f->d_xy);
        if (f->own_accum && f->d_accum) (void)hipFree(f->d_accum);
        delete f;
        return fail(RTMI_ERR_DEVICE, std::string("film allocation: ") + hipGetErrorString(e));
    }
    dev->refs.fetch_add(1);
    *out = f;
    return RTMI_OK;
}

int rtmi_film_create_frame(rtmi_device *dev, uint32_t offset_x, uint32_t offset_y, uint32_t width, uint32_t height,
                           rtmi_film **out) {
    if (!dev || !out) return fail(RTMI_ERR_ARG, "rtmi_film_create_frame: NULL argument");
    if ((uint64_t)width * height >= (1ull << 30)) return fail(RTMI_ERR_LIMIT, "2^30 pixels or more in one film");
    std::vector<uint32_t> xy((size_t)width * height * 2);
    size_t k = 0;
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            xy[k++] = offset_x + x;
            xy[k++] = offset_y + y;
        }
    if (int rc = rtmi_film_create(dev, xy.data(), xy.size() / 2, nullptr, out)) return rc;
    (*out)->is_frame = true;
    (*out)->frame_x = offset_x;
    (*out)->frame_y = offset_y;
    (*out)->frame_w = width;
    (*out)->frame_h = height;
    return RTMI_OK;
}

int rtmi_film_create_tiles(rtmi_device *dev, uint32_t offset_x, uint32_t offset_y, uint32_t width, uint32_t height,
                           uint32_t grain, uint32_t first_tile, uint32_t tile_stride, uint32_t block_w, uint32_t block_h,
                           rtmi_film **out) {
    if (!dev || !out) return fail(RTMI_ERR_ARG, "rtmi_film_create_tiles: NULL argument");
    if (grain == 0 || tile_stride == 0) return fail(RTMI_ERR_ARG, "rtmi_film_create_tiles: grain and stride must be > 0");
    if ((block_w == 0) != (block_h == 0)) return fail(RTMI_ERR_ARG, "rtmi_film_create_tiles: give both block sides or none");
    const uint32_t n_tiles = rtmi_tile_count(width, height, grain);
    std::vector<uint32_t> xy;
    for (uint32_t t = first_tile; t < n_tiles; t += tile_stride) {
        uint32_t r[4];
        if (int rc = rtmi_tile_get(offset_x, offset_y, width, height, grain, t, r)) return rc;
        const uint32_t bw = block_w ? block_w : r[1] - r[0], bh = block_h ? block_h : 1;
        for (uint32_t y0 = r[2]; y0 < r[3]; y0 += bh)
            for (uint32_t x0 = r[0]; x0 < r[1]; x0 += bw)
                for (uint32_t y = y0; y < std::min(y0 + bh, r[3]); ++y)
                    for (uint32_t x = x0; x < std::min(x0 + bw, r[1]); ++x) {
                        xy.push_back(x);
                        xy.push_back(y);
                    }
        if (t > UINT32_MAX - tile_stride) break;
    }
    return rtmi_film_create(dev, xy.data(), xy.size() / 2, nullptr, out);
}

size_t rtmi_film_pixel_count(const rtmi_film *film) { return film ? film->n_pixels : 0; }

int rtmi_film_pixels(rtmi_film *film, uint32_t *xy) {
    if (!film || (film->n_pixels && !xy)) return fail(RTMI_ERR_ARG, "rtmi_film_pixels: NULL argument");
    if (film->n_pixels == 0) return RTMI_OK;
    std::lock_guard<std::mutex> g(film->dev->lock);
    USE_DEVICE(film->dev);
    HIP_TRY(hipMemcpy(xy, film->d_xy, film->n_pixels * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

void rtmi_film_destroy(rtmi_film *film) {
    if (!film) return;
    DeviceScope scope(film->dev->ordinal);
    (void)hipStreamSynchronize(film->dev->stream);
    if (film->d_xy) (void)hipFree(film->d_xy);
    if (film->own_accum && film->d_accum) (void)hipFree(film->d_accum);
    if (film->d_state) (void)hipFree(film->d_state);
    if (film->d_hits) (void)hipFree(film->d_hits);
    if (film->d_queue[0]) (void)hipFree(film->d_queue[0]);
    if (film->d_queue[1]) (void)hipFree(film->d_queue[1]);
    if (film->d_counts) (void)hipFree(film->d_counts);
    if (film->d_resolve) (void)hipFree(film->d_resolve);
    if (film->d_done) (void)hipFree(film->d_done);
    rtmi_gather_plan_free(film);
    rtmi_device *dev = film->dev;
    delete film;
    rtmi_device_release(dev);
}

int rtmi_film_clear(rtmi_film *film) {
    if (!film) return fail(RTMI_ERR_ARG, "film is NULL");
    std::lock_guard<std::mutex> g(film->dev->lock);
    USE_DEVICE(film->dev);
    HIP_TRY(hipMemsetAsync(film->d_accum, 0, film->n_pixels * RTMI_ACCUM_PLANES * sizeof(float), film->dev->stream));
    HIP_TRY(hipStreamSynchronize(film->dev->stream));
    return RTMI_OK;
}

float *rtmi_film_accum_device(rtmi_film *film) { return film ? film->d_accum : nullptr; }

int rtmi_film_download(rtmi_film *film, float *accum) {
    if (!film || !accum) return fail(RTMI_ERR_ARG, "rtmi_film_download: NULL argument");
    std::lock_guard<std::mutex> g(film->dev->lock);
    USE_DEVICE(film->dev);
    HIP_TRY(hipStreamSynchronize(film->dev->stream));
    HIP_TRY(hipMemcpy(accum, film->d_accum, film->n_pixels * RTMI_ACCUM_PLANES * sizeof(float),
                      hipMemcpyDeviceToHost));
    return RTMI_OK;
}

int rtmi_film_resolve(rtmi_film *film, float *out, int srgb) {
    if (!film || !out) return fail(RTMI_ERR_ARG, "rtmi_film_resolve: NULL argument");
    if (film->n_pixels == 0) return RTMI_OK;
    rtmi_device *dev = film->dev;
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    const size_t bytes = film->n_pixels * RTMI_RESOLVE_PLANES * sizeof(float);
    if (!film->d_resolve) HIP_TRY(hipMalloc((void **)&film->d_resolve, bytes));
    if (rtk::launch_resolve(film->d_accum, (uint32_t)film->n_pixels, film->d_resolve, srgb, dev->stream))
        return fail(RTMI_ERR_DEVICE, "resolve kernel launch failed");
    HIP_TRY(hipMemcpyAsync(out, film->d_resolve, bytes, hipMemcpyDeviceToHost, dev->stream));
    HIP_TRY(hipStreamSynchronize(dev->stream));
    return RTMI_OK;
}

int rtmi_render(rtmi_scene *scene, rtmi_film *film, const rtmi_camera *camera, const rtmi_render_params *params,
                rtmi_render_stats *stats) {
    if (!scene || !film || !camera || !params) return fail(RTMI_ERR_ARG, "rtmi_render: NULL argument");
    if (!scene->committed) return fail(RTMI_ERR_STATE, "scene not committed");
    if (scene->dev != film->dev) return fail(RTMI_ERR_ARG, "scene and film belong to different devices");
    if (params->sample_end < params->sample_begin) return fail(RTMI_ERR_ARG, "sample range is reversed");
    if (params->spp == 0) return fail(RTMI_ERR_ARG, "spp must be at least 1");
    if (params->integrator > RTMI_INTEGRATOR_RANDOM_WALK) return fail(RTMI_ERR_ARG, "unknown integrator");
    if ((params->flags & ~0x33u) != 0 || (params->flags & 3u) == 3u || ((params->flags >> 4) & 3u) == 3u)
        return fail(RTMI_ERR_ARG, "unknown sampler or filter kind in flags");
    if (camera->width == 0 || camera->height == 0) return fail(RTMI_ERR_ARG, "camera has no pixels");
    if (stats) memset(stats, 0, sizeof *stats);
    const uint32_t n_samples = params->sample_end - params->sample_begin;
    const size_t npix = film->n_pixels;
    if (n_samples == 0 || npix == 0 || params->max_depth == 0) return RTMI_OK; // nothing accumulates
    if (params->max_depth > RTMI_MAX_RAY_DEPTH) return fail(RTMI_ERR_LIMIT, "max_depth above RTMI_MAX_RAY_DEPTH");

    rtmi_device *dev = scene->dev;
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    hipStream_t stream = dev->stream;
    const bool adaptive = params->allowed_error > 0.0f;

    // path ids are laid out per 64-pixel block (see kernels.h): pad the pixel count to whole blocks
    const size_t n_blocks = (npix + rtk::PIXEL_BLOCK - 1) / rtk::PIXEL_BLOCK;
    const size_t padded_pix = n_blocks * rtk::PIXEL_BLOCK;
    const size_t seg_blocks = (n_blocks + rtk::SEGMENTS - 1) / rtk::SEGMENTS;
    const size_t queue_pix = seg_blocks * rtk::SEGMENTS * rtk::PIXEL_BLOCK; // per sample, all segments
    // a plane of path or queue data is addressed with a 32-bit byte offset: one sample of every pixel must fit
    if (queue_pix >= (1ull << 30)) return fail(RTMI_ERR_LIMIT, "film has 2^30 pixels or more");

    // samples traced together per pass
    // Default: as many samples of every pixel in flight as fit a quarter of the free HBM (164 bytes of state, hit
    // record and two queue entries per path), so that the long, thin tail of deep bounces is paid once per frame and
    // not once per small pass.  With --allowed-error a pixel that has become precise is not traced in later passes of
    // the call, so the default is short passes: at most 7 samples per pixel and call are traced without being used.
    uint32_t spp_pass = params->samples_per_pass;
    if (spp_pass == 0) {
        size_t free_b = 0, total_b = 0;
        size_t target_paths = (size_t)8 << 20;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t per_path = (rtk::STATE_PLANES + rtk::HIT_PLANES + 2 * rtk::PAYLOAD_PLANES) * sizeof(float);
            target_paths = std::max(target_paths, (free_b + film->cap_paths * per_path) / 4 / per_path);
        }
        target_paths = std::min<size_t>(target_paths, (size_t)1 << 28);
        spp_pass = (uint32_t)std::max<size_t>(1, target_paths / npix);
        if (adaptive) spp_pass = std::min(spp_pass, 8u);
    }
    spp_pass = std::min(spp_pass, n_samples);
    if ((size_t)spp_pass * queue_pix >= (1ull << 30))
        spp_pass = (uint32_t)std::max<size_t>(1, ((1ull << 30) - 1) / queue_pix);
    const size_t max_paths = (size_t)spp_pass * queue_pix;
    const uint32_t n_passes = (n_samples + spp_pass - 1) / spp_pass;
    // counters of one pass (reused by the next): SEGMENTS per bounce, each on its own 256-byte line
    const size_t per_bounce = (size_t)rtk::SEGMENTS * rtk::COUNTER_STRIDE;
    const size_t n_counters = ((size_t)params->max_depth + 2) * per_bounce;

    if (film->cap_paths < max_paths) {
        if (film->d_state) (void)hipFree(film->d_state);
        if (film->d_hits) (void)hipFree(film->d_hits);
        if (film->d_queue[0]) (void)hipFree(film->d_queue[0]);
        if (film->d_queue[1]) (void)hipFree(film->d_queue[1]);
        film->d_state = film->d_hits = nullptr;
        film->d_queue[0] = film->d_queue[1] = nullptr;
        film->cap_paths = 0;
        HIP_TRY(hipMalloc((void **)&film->d_state, max_paths * rtk::STATE_PLANES * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&film->d_hits, max_paths * rtk::HIT_PLANES * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&film->d_queue[0], max_paths * rtk::PAYLOAD_PLANES * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&film->d_queue[1], max_paths * rtk::PAYLOAD_PLANES * sizeof(float)));
        film->cap_paths = max_paths;
    }
    if (film->cap_counters < n_counters) {
        if (film->d_counts) (void)hipFree(film->d_counts);
        film->d_counts = nullptr;
        film->cap_counters = 0;
        HIP_TRY(hipMalloc((void **)&film->d_counts, 2 * n_counters * sizeof(uint32_t)));
        film->cap_counters = n_counters;
    }
    // "precise enough, done for this call" flag per pixel + their count: only when a call with --allowed-error spans
    // several passes (within one pass accumulate_kernel's own sample loop stops the pixel)
    const bool use_done = adaptive && n_passes > 1;
    if (use_done) {
        if (!film->d_done) HIP_TRY(hipMalloc((void **)&film->d_done, (npix + 1) * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(film->d_done, 0, (npix + 1) * sizeof(uint32_t), stream));
    }
    uint32_t *d_counts = film->d_counts;
    uint32_t *d_work = film->d_counts + film->cap_counters;
    // only the counters of the bounces this call can reach are zeroed per pass
    const size_t used_counters = n_counters;
    std::vector<uint32_t> h_all;

    rtk::RenderArgs a{};
    a.scene = scene->view;
    a.camera.width = camera->width;
    a.camera.height = camera->height;
    a.camera.half_width = camera->half_width;
    a.camera.half_height = camera->half_height;
    a.camera.focal_length = camera->focal_length;
    a.camera.aperture = camera->aperture;
    memcpy(a.camera.lens, camera->lens, 12);
    memcpy(a.camera.rot, camera->rot, 16);
    a.pixels = film->d_xy;
    a.state = film->d_state;
    a.hits = film->d_hits;
    a.payload[0] = film->d_queue[0];
    a.payload[1] = film->d_queue[1];
    a.visit_counters = dev->count_visits ? dev->d_visits : nullptr;
    a.pixel_done = use_done ? film->d_done : nullptr;
    a.n_pixels = (uint32_t)npix;
    a.n_blocks = (uint32_t)n_blocks;
    a.strata = (uint32_t)floorf(sqrtf((float)params->spp)); // executor.rs:218
    if (a.strata == 0) a.strata = 1;
    a.flags = params->flags;
    a.max_depth = params->max_depth;
    a.integrator = params->integrator;
    a.seed = params->seed;

    a.allowed_error = adaptive ? params->allowed_error : 0.0f;
    a.refill_below = 40; // swept 32 .. 56 on 5 M triangles: flat from 32 to 40, 3 % slower on deep bounces at 48
    a.leaf_bias = 0;
    if (const char *v = getenv("RTMI_REFILL_BELOW")) a.refill_below = atoi(v);
    if (const char *v = getenv("RTMI_LEAF_BIAS")) a.leaf_bias = atoi(v);
    // the first bounces (rays that still share the top of the tree in L2) are bound by instruction issue and gain from
    // deferring the record tests (bounce 2: 8 % since a node is three loads); on deeper bounces the two kernels are
    // level, and those keep the voting kernel
    // swept 8 .. 64 on 5 M triangles after the node fetch shrank: 32 .. 64 level, 16 slower by 0.7 %; on a scene that
    // fits L1 (a handful of nodes per ray, every path in the tail kernel) 32 is 7 % slower than 16
    int queue_low = scene->info.node_bytes + scene->info.prim_bytes <= (32u << 10) ? 16 : 32;
    uint32_t queue_depths = 3;
    if (const char *v = getenv("RTMI_QUEUE_LOW")) queue_low = std::max(0, std::min(64, atoi(v)));
    if (const char *v = getenv("RTMI_QUEUE_DEPTHS")) queue_depths = (uint32_t)std::max(0, atoi(v));
    int tail_refill = 64; // lanes with a path below which a wave of the tail kernel takes new paths (64: as soon as one is idle)
    if (const char *v = getenv("RTMI_TAIL_REFILL")) tail_refill = std::max(1, std::min(64, atoi(v)));
    int tail_shade = 32; // lanes waiting for shading that make a wave of the tail kernel shade (0: when they are the majority)
    if (const char *v = getenv("RTMI_TAIL_SHADE")) tail_shade = std::max(0, std::min(64, atoi(v)));
    int queue_refill = 48; // refill threshold of that kernel (lanes still traversing or waiting for queued tests)
    if (const char *v = getenv("RTMI_QUEUE_REFILL")) queue_refill = std::max(1, std::min(64, atoi(v)));
    a.queue_low = 0;
    a.refill_below = std::max(1, std::min(64, a.refill_below)); // below 1 a drained wave would never flush
    const int vote_refill = a.refill_below;
    a.spill = scene->d_spill;

    // the copy stream, its pinned slots and events are made once per device; a failure half-way is retried from the
    // piece that is missing
    if (!dev->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&dev->copy_stream, hipStreamNonBlocking));
    if (!dev->h_counts)
        HIP_TRY(hipHostMalloc((void **)&dev->h_counts, rtmi_device::COUNT_RING * per_bounce * sizeof(uint32_t),
                              hipHostMallocDefault));
    for (int r = 0; r < rtmi_device::COUNT_RING; ++r) {
        if (!dev->ev_shaded[r]) HIP_TRY(hipEventCreateWithFlags(&dev->ev_shaded[r], hipEventDisableTiming));
        if (!dev->ev_copied[r]) HIP_TRY(hipEventCreateWithFlags(&dev->ev_copied[r], hipEventDisableTiming));
    }
    Stopwatch watch;
    float total_ms = 0.0f, bounce_ms = 0.0f, extend_ms = 0.0f;
    // per-bounce view of the same launches (rtmi_device_last_render_profile); sized once, the stopwatch keeps pointers
    struct BounceTimes {
        float extend_ms[2] = {0.0f, 0.0f}; // by kernel: voting, queued
        float shade_ms = 0.0f, tail_ms = 0.0f;
        uint32_t launches[2] = {0, 0}, tail_launches = 0;
        uint64_t rays = 0, tail_rays = 0;
    };
    std::vector<BounceTimes> prof((size_t)params->max_depth + 1);
    const int t_total = watch.start(stream, &total_ms);
    if (t_total < 0) return fail(RTMI_ERR_DEVICE, "hipEventCreate failed");
    int rc = RTMI_OK;
    uint32_t sample = params->sample_begin;
    uint64_t rays = 0, paths = 0, tail_rays_traced = 0;
    uint64_t tail_rays = 1u << 25;
    // A scene that fits a CU's L1 (the Cornell box: 4 KB) costs a few fetches per ray that never leave the chip; what
    // a bounce of the wavefront pair then mostly pays for is streaming ray, hit and queue planes through HBM, which the
    // tail kernel does not do (Cornell box 1920x1080 100 spp: 168.9 -> 149.3 ms with the tail kernel from bounce 1;
    // no difference on a 0.5 MB scene).
    bool tail_auto = true; // the threshold follows the size of the pass (below)
    if (scene->info.node_bytes + scene->info.prim_bytes <= (32u << 10)) tail_rays = UINT64_MAX, tail_auto = false;
    if (const char *v = getenv("RTMI_TAIL_RAYS")) tail_rays = (uint64_t)atoll(v), tail_auto = false;
    uint32_t deepest = 0, launches = 0;
    uint32_t n_done = 0; // pixels already precise enough for this call (use_done only)
    const uint32_t waves_per_block = rtk::BLOCK / 64;
    // Synchronises with the stream after a small device-to-host copy; every timed interval closed before it is then
    // finished.
    auto read_back = [&](void *dst, const void *src, size_t bytes) -> int {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return fail(RTMI_ERR_DEVICE, std::string("render: ") + hipGetErrorString(e));
        watch.collect();
        return RTMI_OK;
    };
    for (uint32_t pass = 0; pass < n_passes && rc == RTMI_OK; ++pass) {
        const uint32_t k = std::min(spp_pass, params->sample_end - sample);
        a.samples = k;
        a.n_paths = (uint32_t)((size_t)k * padded_pix);
        a.seg_cap = (uint32_t)(seg_blocks * rtk::PIXEL_BLOCK * k);
        a.sample_base = sample;
        a.counts = d_counts;
        a.work = d_work;
        hipError_t e = hipMemsetAsync(d_counts, 0, used_counters * sizeof(uint32_t), stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_work, 0, used_counters * sizeof(uint32_t), stream);
        if (e != hipSuccess) {
            rc = fail(RTMI_ERR_DEVICE, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
            break;
        }
        const int t_pass = watch.start(stream, &bounce_ms);
        if (rtk::launch_raygen(a, stream)) {
            rc = fail(RTMI_ERR_DEVICE, "raygen kernel launch failed");
            break;
        }
        // `live` / `longest`: rays entering a bounce and the entries of its longest segment.  For bounce 0 they are
        // known; from then on they are the survivor counters the shade kernel leaves, copied to the host on the copy
        // stream while the render stream goes on.  The host stays one bounce ahead of the device: bounce d is launched
        // with the counts of bounce d - 1 as upper bounds (survivors never outnumber the rays they come from; the
        // kernels read the exact counts on the device), so no bounce waits for the host.  Once fewer than tail_rays
        // are left a launch per bounce costs more than the bounce, and one tail_kernel launch runs the remaining
        // paths to their end.
        uint64_t live = (uint64_t)k * (npix - n_done);
        uint32_t longest = a.seg_cap;
        uint32_t launched = 0, tail_from = 0; // bounces whose counters are valid; first bounce of the tail (0 = none)
        paths += live;
        const uint64_t camera_rays = live;
        uint64_t expected = live; // rays expected to enter the bounce: the bound scaled by the last bounce's survival rate
        // The tail kernel costs about 3 ms (its deepest path) + 0.31 ns per ray whatever it is given; the wavefront pair
        // costs 0.23 .. 0.4 ns per ray, more the deeper the bounce and the fewer its rays.  On a full 1080p frame of 100
        // samples the switch pays from about a sixth of the camera rays on (2^25 rays); a pass an eighth of that size (one
        // rank's share of an 8-GPU job) would then hand everything after bounce 0 to the tail kernel, and does better
        // keeping bounce 1 (swept 2^21 .. 2^25 on such a share: best at 6-8 M).
        const uint64_t pass_tail_rays =
            tail_auto ? std::min<uint64_t>(tail_rays, std::max<uint64_t>(6000000u, (uint64_t)(0.16 * (double)camera_rays))) : tail_rays;
        for (uint32_t depth = 0; depth < params->max_depth && live > 0; ++depth) {
            if (depth >= 2) {
                const uint64_t before = live;
                // exact input of bounce depth - 1 (copied after bounce depth - 2 was shaded): the bound for this bounce
                const int slot = (int)((depth - 1) % rtmi_device::COUNT_RING);
                hipError_t ce = hipEventSynchronize(dev->ev_copied[slot]);
                if (ce != hipSuccess) {
                    rc = fail(RTMI_ERR_DEVICE, std::string("render: ") + hipGetErrorString(ce));
                    break;
                }
                watch.collect();
                const uint32_t *hc = dev->h_counts + (size_t)slot * per_bounce;
                live = 0;
                longest = 0;
                for (int x = 0; x < rtk::SEGMENTS; ++x) {
                    live += hc[(size_t)x * rtk::COUNTER_STRIDE];
                    longest = std::max(longest, hc[(size_t)x * rtk::COUNTER_STRIDE]);
                }
                if (live == 0) break; // bounce depth - 1 already ran on an empty queue
                expected = (uint64_t)((double)live * (double)live / (double)std::max<uint64_t>(before, 1));
            }
            if (depth > 0 && expected < pass_tail_rays) {
                // persistent waves: a resident grid at most, one lane per path when there are fewer paths than that
                const uint64_t want = (live + rtk::BLOCK - 1) / rtk::BLOCK;
                a.claim = 64u;
                a.refill_below = tail_refill;
                a.tail_shade = tail_shade;
                a.queue_low = queue_low;
                const int t_tail = watch.start(stream, &prof[depth].tail_ms);
                if (rtk::launch_tail(a, depth, (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(scene->grid_blocks_tail, want)), stream))
                    rc = fail(RTMI_ERR_DEVICE, "tail kernel launch failed");
                watch.stop(t_tail, stream);
                ++prof[depth].tail_launches;
                tail_from = depth;
                launched = params->max_depth;
                break;
            }
            // grid sized to the work: one wave per 64 queue entries, capped at the resident grid; a wave claims
            // several batches per atomic when the queue is long
            a.queue_low = depth < queue_depths ? queue_low : 0;
            const uint32_t resident = a.queue_low > 0 && a.visit_counters == nullptr ? scene->grid_blocks_queue : scene->grid_blocks_vote;
            const uint64_t entries = depth == 0 ? (uint64_t)a.n_paths : live;
            const uint64_t want_blocks = (entries + 64 * waves_per_block - 1) / (64 * waves_per_block);
            const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(resident, want_blocks));
            const uint64_t per_wave = entries / ((uint64_t)grid * waves_per_block * rtk::SEGMENTS) / 8;
            a.claim = 64u * (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(16, per_wave / 64));
            // timed on its own: the traversal launch is what the roofline of the dominant kernel needs
            const int t_extend = watch.start(stream, &extend_ms);
            a.refill_below = a.queue_low > 0 ? queue_refill : vote_refill;
            const int which = a.queue_low > 0 && a.visit_counters == nullptr ? 1 : 0;
            const int t_extend_d = watch.start(stream, &prof[depth].extend_ms[which]);
            int lrc = rtk::launch_extend(a, depth, grid, stream);
            watch.stop(t_extend, stream);
            watch.stop(t_extend_d, stream);
            ++prof[depth].launches[which];
            const int t_shade = watch.start(stream, &prof[depth].shade_ms);
            if (!lrc) lrc = rtk::launch_shade(a, depth, longest, stream);
            watch.stop(t_shade, stream);
            if (lrc) {
                rc = fail(RTMI_ERR_DEVICE, "extend or shade kernel launch failed");
                break;
            }
            ++launches;
            ++launched;
            if (depth + 1 == params->max_depth) break;
            // the survivors' counters, to the host behind the render stream's back
            {
                const int slot = (int)((depth + 1) % rtmi_device::COUNT_RING);
                hipError_t ce = hipEventRecord(dev->ev_shaded[slot], stream);
                if (ce == hipSuccess) ce = hipStreamWaitEvent(dev->copy_stream, dev->ev_shaded[slot], 0);
                if (ce == hipSuccess)
                    ce = hipMemcpyAsync(dev->h_counts + (size_t)slot * per_bounce, d_counts + (size_t)(depth + 1) * per_bounce,
                                        per_bounce * sizeof(uint32_t), hipMemcpyDeviceToHost, dev->copy_stream);
                if (ce == hipSuccess) ce = hipEventRecord(dev->ev_copied[slot], dev->copy_stream);
                if (ce != hipSuccess) {
                    rc = fail(RTMI_ERR_DEVICE, std::string("render: ") + hipGetErrorString(ce));
                    break;
                }
            }
        }
        if (rc != RTMI_OK) break;
        watch.stop(t_pass, stream);
        // queued before the host reads the pass's counters, so that the device does not wait for that read
        if (rtk::launch_accumulate(a, film->d_accum, 1.0f, stream)) {
            rc = fail(RTMI_ERR_DEVICE, "accumulate kernel launch failed");
            break;
        }
        // exact ray count and deepest bounce of the pass from the counters of every bounce that ran (read in chunks
        // of 256 bounces up to the first empty one: live counts never grow from one bounce to the next)
        rays += camera_rays;
        prof[0].rays += camera_rays;
        bool ended = false;
        for (uint32_t d0 = 1; d0 < launched && !ended; d0 += 256) {
            const uint32_t d1 = std::min(launched, d0 + 256);
            h_all.resize((size_t)(d1 - d0) * per_bounce);
            if ((rc = read_back(h_all.data(), d_counts + (size_t)d0 * per_bounce, h_all.size() * sizeof(uint32_t))) !=
                RTMI_OK)
                break;
            for (uint32_t d = d0; d < d1; ++d) {
                uint64_t n_d = 0;
                for (int x = 0; x < rtk::SEGMENTS; ++x)
                    n_d += h_all[(size_t)(d - d0) * per_bounce + (size_t)x * rtk::COUNTER_STRIDE];
                if (n_d == 0) {
                    ended = true;
                    break;
                }
                rays += n_d;
                if (tail_from != 0 && d >= tail_from) {
                    tail_rays_traced += n_d;
                    prof[tail_from].tail_rays += n_d;
                } else {
                    prof[d].rays += n_d;
                }
                if (d > deepest) deepest = d;
            }
        }
        if (rc != RTMI_OK) break;
        if (use_done && pass + 1 < n_passes) rc = read_back(&n_done, film->d_done + npix, sizeof(uint32_t));
        sample += k;
    }
    watch.stop_all(stream);
    hipError_t e = hipStreamSynchronize(stream);
    if (rc == RTMI_OK && e != hipSuccess) rc = fail(RTMI_ERR_DEVICE, std::string("render: ") + hipGetErrorString(e));
    if (e == hipSuccess) watch.collect();
    if (rc == RTMI_OK && e == hipSuccess) {
        dev->profile.clear();
        for (uint32_t d = 0; d < prof.size(); ++d) {
            const BounceTimes &b = prof[d];
            for (uint32_t k = 0; k < 2; ++k)
                if (b.launches[k]) {
                    // rays and shading time of a bounce traced by both kernels (in different passes) go to the first entry
                    const bool first = k == 0 || b.launches[0] == 0;
                    dev->profile.push_back(rtmi_bounce_profile{d, k, first ? b.rays : 0, b.extend_ms[k], first ? b.shade_ms : 0.0f,
                                                               b.launches[k], 0});
                }
            if (b.tail_launches)
                dev->profile.push_back(rtmi_bounce_profile{d, RTMI_KERNEL_TAIL, b.tail_rays, b.tail_ms, 0.0f, b.tail_launches, 0});
        }
    }
    if (rc == RTMI_OK && stats) {
        stats->rays = rays;
        stats->tail_rays = tail_rays_traced;
        stats->paths = paths;
        stats->max_bounces = deepest;
        stats->passes = n_passes;
        stats->bounce_launches = launches;
        stats->bounce_ms = bounce_ms;
        stats->extend_ms = extend_ms;
        stats->gpu_ms = total_ms;
    }
    return rc;
}

static uint32_t div_ceil_u32(uint32_t x, uint32_t y) { return x / y + (x % y != 0 ? 1u : 0u); }

uint32_t rtmi_tile_count(uint32_t width, uint32_t height, uint32_t grain) {
    if (grain == 0) return 0;
    return div_ceil_u32(width, grain) * div_ceil_u32(height, grain);
}

int rtmi_tile_get(uint32_t offset_x, uint32_t offset_y, uint32_t width, uint32_t height, uint32_t grain,
                  uint32_t idx, uint32_t out[4]) {
    if (!out || grain == 0) return fail(RTMI_ERR_ARG, "rtmi_tile_get: bad argument");
    if (idx >= rtmi_tile_count(width, height, grain)) return fail(RTMI_ERR_ARG, "tile index out of range");
    const uint32_t cols = div_ceil_u32(width, grain);
    const uint32_t x = idx % cols, y = idx / cols;
    out[0] = offset_x + x * grain;
    out[1] = offset_x + std::min(width, (x + 1) * grain);
    out[2] = offset_y + y * grain;
    out[3] = offset_y + std::min(height, (y + 1) * grain);
    return RTMI_OK;
}

int rtmi_tile_partition(uint32_t n_tiles, uint32_t world, uint32_t rank, uint32_t *out_first, uint32_t *out_stride,
                        uint32_t *out_count) {
    if (world == 0 || rank >= world) return fail(RTMI_ERR_ARG, "rtmi_tile_partition: rank must be below world");
    if (out_first) *out_first = rank;
    if (out_stride) *out_stride = world;
    if (out_count) *out_count = rank < n_tiles ? (n_tiles - rank + world - 1) / world : 0;
    return RTMI_OK;
}

} // extern "C"
