/*
 * rtmi.h -- C ABI of the MI355X path-tracing core.
 *
 * One shared library (librtmi.so, HIP/gfx950) replaces the reference's only accelerator back end: the Embree-backed
 * scene (crates/rt/src/aggregate/embree.rs) and the CPU tile executor that drives it
 * (crates/rt-cli/src/executor.rs).  Every entry point names the reference interface it stands in for.  All
 * compute runs on the GPU; there is no CPU fallback, and every call fails with RTMI_ERR_DEVICE when no device is
 * usable.
 *
 * Conventions
 *   - Plain C types only.  Handles are opaque and owned by the caller; release them with the matching *_destroy.
 *   - Input arrays are borrowed for the duration of the call and copied (as embree.rs:214-241 copies into
 *     Embree-owned buffers).  Output arrays are caller-allocated.
 *   - Functions returning int return RTMI_OK (0) or a negative RTMI_ERR_*; rtmi_last_error() then gives a
 *     message for the calling thread.  No exception crosses this boundary.  (The reference panics on geometry
 *     errors, embree.rs:190-248, and returns anyhow::Result from commit, embree.rs:63-69; a binding maps non-zero
 *     to those.)
 *   - insert_* / load_obj / commit on one scene must not run concurrently (the reference takes &mut self).
 *     Queries and renders on a committed scene may be issued from several threads (Shape is Sync + Send,
 *     shape/mod.rs:3): rtmi_intersect, rtmi_render and the film calls serialise per device on the library's stream;
 *     rtmi_intersect_device launches on the caller's stream and may overlap with anything.
 *   - Every entry point makes its device current while it runs and restores the calling thread's current HIP device
 *     before it returns.
 *   - A scene is immutable once committed (the reference enforces it by borrow, embree.rs:81-84); inserting
 *     after commit returns RTMI_ERR_STATE.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_OK 0
#define RTMI_ERR_ARG (-1)    /* bad argument */
#define RTMI_ERR_DEVICE (-2) /* no usable HIP device, or a HIP call failed */
#define RTMI_ERR_STATE (-3)  /* call not valid in the object's state (e.g. insert after commit) */
#define RTMI_ERR_IO (-4)     /* file could not be read or parsed */
#define RTMI_ERR_CANCELLED (-5) /* the commit progress callback returned 0 */
#define RTMI_ERR_LIMIT (-6)  /* a size exceeds what the kernels support */

#define RTMI_INVALID_ID 0xFFFFFFFFu
#define RTMI_MAX_RAY_DEPTH 4096u /* largest rtmi_render_params.max_depth (the reference's default is 64) */

/* BxDF kinds: crates/rt/src/material/mod.rs:98-101 (Diffuse), :133-137 (Dielectric), :315-318 (ThinDielectric),
 * :369-372 (Emit). */
enum { RTMI_MAT_DIFFUSE = 0, RTMI_MAT_DIELECTRIC = 1, RTMI_MAT_THIN_DIELECTRIC = 2, RTMI_MAT_EMIT = 3 };

/* crates/rt-cli/src/utils/mod.rs:140-155: `-i path-tracer` (default) / `-i basic`. */
enum { RTMI_INTEGRATOR_PATH_TRACER = 0, RTMI_INTEGRATOR_RANDOM_WALK = 1 };

/* MaterialDescriptor's BxDF as a tagged POD (material/mod.rs:401-404). rgb = albedo (Diffuse) or le (Emit). */
typedef struct rtmi_material {
    uint32_t kind;
    float rgb[3];
    float ior;
    float roughness;
    uint32_t pad[2];
} rtmi_material;

/* Ray, crates/rt/src/ray.rs:8-12; the layout Embree's RTCRay starts with (embree.rs:102-112). */
typedef struct rtmi_ray {
    float org[3];
    float tnear;
    float dir[3];
    float tfar;
} rtmi_ray;

/* RayIntersection<local_info::Full>, crates/rt/src/shape/mod.rs:22-41, plus the ids Embree reports
 * (embree.rs:129-135).  A miss has geom_id == prim_id == RTMI_INVALID_ID, t = +inf, everything else 0. */
typedef struct rtmi_hit {
    float t;
    float pos[3];
    float n[3];
    float uv[2];
    uint32_t geom_id;
    uint32_t prim_id;
    uint32_t mat_id;
} rtmi_hit;

/* Camera after Camera::new, crates/rt/src/camera.rs:9-57. rot is a quaternion (x, y, z, w). */
typedef struct rtmi_camera {
    uint32_t width, height;
    float half_width, half_height;
    float focal_length;
    float aperture;
    float lens[3];
    float rot[4];
} rtmi_camera;

/* Transform, crates/rt/src/math/transform.rs:6-10. */
typedef struct rtmi_transform {
    float translation[3];
    float scale[3];
    float rot[4];
} rtmi_transform;

/* What Executor carries (crates/rt-cli/src/executor.rs:46-58) plus the sample range (utils/mod.rs:89-97). */
typedef struct rtmi_render_params {
    uint32_t spp;          /* --spp: fixes the strata grid floor(sqrt(spp))^2 (executor.rs:218) */
    uint32_t sample_begin; /* --sample-range, half open */
    uint32_t sample_end;
    uint32_t max_depth;    /* --max-ray-depth (64, utils/mod.rs:149) */
    uint32_t integrator;   /* RTMI_INTEGRATOR_* */
    uint32_t samples_per_pass; /* samples of every pixel traced together; 0 = library default */
    uint64_t seed;         /* --seed */
    float allowed_error;   /* --allowed-error (executor.rs:244-249): within one call a pixel stops accumulating once
                            * the 95 % confidence error of all three colour channels is below this; <= 0 = off.  The
                            * reference applies it per batch of 32 samples: call once per batch to reproduce that.
                            * However many passes the call is split into, a stopped pixel stays stopped until the call
                            * returns, and later passes no longer trace it (with samples_per_pass = 0 such a call uses
                            * passes of at most 8 samples). */
    uint32_t flags;        /* RTMI_SAMPLER_* | RTMI_FILTER_*; 0 = what the executor builds (executor.rs:217-223) */
} rtmi_render_params;

/* Pixel samplers (crates/rt/src/sampler.rs:18-112) and filters (crates/rt/src/filter.rs:11-55, radius 0.7) */
#define RTMI_SAMPLER_STRATIFIED 0x00u
#define RTMI_SAMPLER_UNIFORM 0x01u
#define RTMI_SAMPLER_PIXEL_CENTRE 0x02u /* DummyPixelSampler */
#define RTMI_FILTER_BOX 0x00u
#define RTMI_FILTER_TRIANGLE 0x10u
#define RTMI_FILTER_NONE 0x20u /* DummyFilter */

typedef struct rtmi_render_stats {
    uint64_t rays;   /* closest-hit queries traced, bounces included */
    uint64_t paths;  /* camera samples started */
    uint32_t passes; /* wavefront passes run */
    uint32_t max_bounces; /* deepest bounce that still had live paths */
    float gpu_ms;    /* device time of the call, from HIP events on the render stream */
    float bounce_ms; /* of which: from the camera-ray kernel to the last bounce's kernels of every pass */
    uint32_t bounce_launches; /* closest-hit (traversal) kernel launches issued, one per bounce and pass */
    float extend_ms; /* device time inside those closest-hit kernel launches, from HIP events around each launch */
    uint64_t tail_rays; /* of `rays`: traced by the run-to-completion kernel that finishes the last few paths of a pass */
} rtmi_render_stats;

/* Where the device time of the last rtmi_render on a device went, bounce by bounce (rtmi_device_last_render_profile).
 * The reference times one scope, the whole tile phase (rt-cli/src/renderer.rs:65); it has no per-bounce view. */
enum { RTMI_KERNEL_EXTEND = 0, RTMI_KERNEL_EXTEND_QUEUE = 1, RTMI_KERNEL_TAIL = 2 };
typedef struct rtmi_bounce_profile {
    uint32_t depth;    /* bounce; for RTMI_KERNEL_TAIL the first bounce the run-to-completion kernel took over */
    uint32_t kernel;   /* RTMI_KERNEL_*: which closest-hit kernel traced the bounce */
    uint64_t rays;     /* rays traced at this bounce (RTMI_KERNEL_TAIL: at this and every deeper bounce), all passes */
    float extend_ms;   /* device time inside the closest-hit launches (RTMI_KERNEL_TAIL: inside the tail launches) */
    float shade_ms;    /* device time inside the shading launches of the bounce (0 for the tail, which shades inside) */
    uint32_t launches; /* closest-hit launches added up in this entry (one per pass that reached the bounce) */
    uint32_t reserved;
} rtmi_bounce_profile;

typedef struct rtmi_scene_info {
    uint64_t n_prims;
    uint64_t n_nodes;
    uint64_t node_bytes;
    uint64_t prim_bytes;
    uint32_t n_geoms;
    uint32_t n_materials;
    uint32_t stack_need; /* traversal stack entries the committed tree can require */
    uint32_t max_depth;  /* depth of the wide tree */
    float build_ms;      /* host time of commit (tree build + flatten + upload) */
    float sah_cost;
    uint32_t n_records;      /* leaf records in the tree: one or two triangles or a sphere each; a record the builder cut
                                at a plane (spatial split) is held once per part */
    uint32_t spatial_splits; /* nodes of the binary build that cut records at a plane instead of sorting them into two
                                sets (0 for a plain object-split tree; RTMI_SPATIAL=0 in the environment builds one) */
} rtmi_scene_info;

typedef struct rtmi_device rtmi_device;
typedef struct rtmi_scene rtmi_scene;
typedef struct rtmi_film rtmi_film;
typedef struct rtmi_comm rtmi_comm;

const char *rtmi_last_error(void);
const char *rtmi_version(void);

/* embree4_rs Device::try_new (crates/rt-cli/src/main.rs:74-90).  hip_ordinal < 0 selects the current device. */
int rtmi_device_create(int hip_ordinal, rtmi_device **out);
/* Releases the handle.  Scenes and films created on the device keep it alive until they are destroyed too (the
 * reference's Arc-like lifetime: CommittedScene borrows the Device), so the order of destruction is free. */
void rtmi_device_destroy(rtmi_device *dev);

/* The stream the library launches on (a hipStream_t), for callers that time or order work against it. */
void *rtmi_device_stream(rtmi_device *dev);
/* Work counters the reference does not have (its only counter is utils/counter.rs:84-102): when enabled, every
 * render adds its BVH node fetches to out[0] and its primitive tests to out[1].  Enabling resets both. */
int rtmi_device_count_visits(rtmi_device *dev, int enable);
int rtmi_device_get_visits(rtmi_device *dev, uint64_t out[2]);
/* Per-bounce device times and ray counts of the last rtmi_render that finished on this device (HIP events around every
 * launch on the library's stream): up to `capacity` entries into out, ordered by bounce; *n_out = entries available.
 * No counterpart in the reference (its only timer is the "run tile renderer" scope, rt-cli/src/renderer.rs:65). */
int rtmi_device_last_render_profile(rtmi_device *dev, rtmi_bounce_profile *out, uint32_t capacity, uint32_t *n_out);

/* EmbreeScene::new (embree.rs:29-51): material 0 is the black "Sky" diffuse. */
int rtmi_scene_create(rtmi_device *dev, rtmi_scene **out);
void rtmi_scene_destroy(rtmi_scene *scene);

/* SceneT::insert_material (scene/mod.rs:11, embree.rs:167-171): returns the MaterialId in *out_id. */
int rtmi_scene_insert_material(rtmi_scene *scene, const rtmi_material *mat, uint32_t *out_id);
/* SceneT::insert_light (scene/mod.rs:12, embree.rs:173-175): stored; no integrator reads lights. */
int rtmi_scene_insert_light(rtmi_scene *scene, const float pos[3]);
/* SceneT::insert_mesh (scene/mod.rs:13-18, embree.rs:177-253): vertices = n_vertices x 3 floats, indices =
 * n_triangles x 3 uint32; returns the GeometryHandle (geometry id) in *out_geom. */
int rtmi_scene_insert_mesh(rtmi_scene *scene, uint32_t material, const float *vertices, size_t n_vertices,
                           const uint32_t *indices, size_t n_triangles, uint32_t *out_geom);
/* SceneT::insert_sphere (scene/mod.rs:20-25, embree.rs:255-265). */
int rtmi_scene_insert_sphere(rtmi_scene *scene, uint32_t material, const float center[3], float radius,
                             uint32_t *out_geom);
/* ObjLoaderExt::load_obj (crates/rt/src/loader/obj.rs:24-123): OBJ + MTL read with tobj's GPU_LOAD_OPTIONS +
 * single_index semantics; every MTL material becomes a Diffuse(Kd); one mesh per model; transform applied to the
 * positions.  *out_first_geom / *out_n_geoms (either may be NULL) receive the range of geometry ids inserted. */
int rtmi_scene_load_obj(rtmi_scene *scene, const char *path, const rtmi_transform *transform,
                        uint32_t default_material, uint32_t *out_first_geom, uint32_t *out_n_geoms);

/* EmbreeScene::commit / commit_with_progress (embree.rs:63-78): builds the SAH BVH on the host, flattens it to
 * 4-wide 64-byte GPU nodes over 64-byte primitive records and uploads them.  The reference asks Embree for
 * RTCBuildQuality::HIGH (embree.rs:33-34), a build with spatial splits; this builder too cuts a record at a plane and
 * keeps it once per part where its cost estimate says that pays (rtmi_scene_info.spatial_splits, n_records).  Hits do
 * not depend on the tree.  progress (may be NULL) is called with values in [0,1]; returning 0 cancels
 * (RTMI_ERR_CANCELLED). */
typedef int (*rtmi_progress_fn)(double amount, void *user);
int rtmi_scene_commit(rtmi_scene *scene, rtmi_progress_fn progress, void *user);

/* Shape::bounding_box (shape/mod.rs:8, embree.rs:146-161). */
int rtmi_scene_bounds(const rtmi_scene *scene, float lo[3], float hi[3]);
int rtmi_scene_get_info(const rtmi_scene *scene, rtmi_scene_info *out);
/* geometry_material map (embree.rs:24,53-61): material of a geometry id. */
int rtmi_scene_geometry_material(const rtmi_scene *scene, uint32_t geom, uint32_t *out_material);

/* Shape::intersection_full (shape/mod.rs:4, embree.rs:101-140) for n rays at once: closest hit in
 * (tnear, tfar].  Host arrays; the rays are copied to the GPU, traced there and the hits copied back, through staging
 * buffers the scene keeps between calls (a call allocates only when n exceeds every earlier n).
 * Ray origins must lie within 16 times the scene's largest |coordinate| of the coordinate origin on every axis: the
 * tree's culling boxes are padded for the rounding of slab distances up to that range, and beyond it a hit could be
 * culled.  A ray outside the range makes the call fail with RTMI_ERR_ARG (start such a ray nearer: move its origin
 * along the direction and shorten tfar). */
int rtmi_intersect(rtmi_scene *scene, const rtmi_ray *rays, rtmi_hit *hits, size_t n);
/* Same with device-resident arrays, asynchronous on hip_stream (a hipStream_t; NULL = the default stream); the
 * origin range above is the caller's to keep (the rays are not inspected).  Launches on different streams may overlap
 * each other and renders: on a tree deeper than the kernels' on-chip stack (rtmi_scene_info.stack_need > 24) each
 * launch takes one of four stack-overflow buffers, created on first use, and a fifth concurrent launch waits on its
 * stream for the first to finish. */
int rtmi_intersect_device(rtmi_scene *scene, const rtmi_ray *d_rays, rtmi_hit *d_hits, size_t n, void *hip_stream);

/* rtmi_intersect_device for callers that cannot allocate device memory while they run (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): no counterpart in the reference.  Same arguments,
 * results and asynchrony as rtmi_intersect_device, but the traversal's stack-overflow buffer is a fixed 32 MiB array
 * inside the library's code object (one copy per GPU, held until the process ends) instead of buffers created on first
 * use, and the call issues no hipMalloc / hipFree, ever.  Limits: trees with rtmi_scene_info.stack_need above
 * RTMI_STATIC_MAX_STACK are refused with RTMI_ERR_LIMIT, and on deep trees the launch uses fewer waves than
 * rtmi_intersect_device so that their overflow fits the array.  Every call on one GPU shares that GPU's array:
 * two of these calls that overlap on one GPU (different streams, or different threads) race on deep trees; keep them
 * on one stream.  The array needs no clearing between calls (a stack entry is written before it is read). */
#define RTMI_STATIC_MAX_STACK 152u
int rtmi_intersect_device_static(rtmi_scene *scene, const rtmi_ray *d_rays, rtmi_hit *d_hits, size_t n,
                                 void *hip_stream);

/* The per-pixel accumulators the executor keeps (Dispatcher::tiles_data, executor.rs:177-205; RaySeries,
 * crates/rt/src/renderer.rs:24-34) for an explicit list of pixels: xy = n_pixels x (x, y) uint32, host array.
 * Accumulators live on the GPU as RTMI_ACCUM_PLANES planes of n_pixels floats, plane-major:
 *   0-2 sum c, 3-5 sum c^2, 6-8 sum w*c, 9 sum w, 10-12 sum normal, 13-15 sum position, 16-18 sum albedo,
 *   19 sum z, 20 sum ray_depth, 21 samples accumulated.
 * d_accum may be a caller-owned device buffer of RTMI_ACCUM_PLANES * n_pixels floats (it is used in place and
 * must outlive the film), or NULL to let the film allocate (and zero) its own. */
#define RTMI_ACCUM_PLANES 22
#define RTMI_RESOLVE_PLANES 15
int rtmi_film_create(rtmi_device *dev, const uint32_t *xy, size_t n_pixels, float *d_accum, rtmi_film **out);
void rtmi_film_destroy(rtmi_film *film);
int rtmi_film_clear(rtmi_film *film);
/* Device pointer of the accumulator planes. */
float *rtmi_film_accum_device(rtmi_film *film);
/* Copies the accumulator planes to the host (RTMI_ACCUM_PLANES * n_pixels floats). */
int rtmi_film_download(rtmi_film *film, float *accum);
/* RaySeries::as_pixelresult (renderer.rs:37-61) on the GPU: out = RTMI_RESOLVE_PLANES planes of n_pixels floats
 * on the host: Normal 0-2, Position 3-5, Albedo 6-8, Color 9-11, Variance 12, Z 13, RayDepth 14.  srgb != 0 also
 * applies OutputBuffers::convert (crates/rt-cli/src/output/mod.rs:46-58): sRGB OETF on the four RGB channels. */
int rtmi_film_resolve(rtmi_film *film, float *out, int srgb);

/* A film over the interleaved share of a frame's tiles one rank of a multi-GPU job renders: tiles first_tile,
 * first_tile + tile_stride, ... of the Tiler (offset, width, height, grain; tile.rs:56-97), tile after tile.  With
 * block_w, block_h > 0 the pixels of a tile are listed block after block of block_w x block_h pixels (row-major
 * inside a block and between blocks) instead of row by row: the renderer traces 64 consecutive pixels of the list per
 * wave, and a compact 8 x 8 block of camera rays stays together in the tree longer than a 32 x 2 strip.  The order
 * never changes a pixel's result. */
int rtmi_film_create_tiles(rtmi_device *dev, uint32_t offset_x, uint32_t offset_y, uint32_t width, uint32_t height,
                           uint32_t grain, uint32_t first_tile, uint32_t tile_stride, uint32_t block_w, uint32_t block_h,
                           rtmi_film **out);
/* A film over every pixel of the window (offset, width, height), row-major: what OutputBuffers holds for the whole
 * image (crates/rt-cli/src/output/mod.rs:13-21) and the destination of rtmi_film_gather. */
int rtmi_film_create_frame(rtmi_device *dev, uint32_t offset_x, uint32_t offset_y, uint32_t width, uint32_t height,
                           rtmi_film **out);
size_t rtmi_film_pixel_count(const rtmi_film *film);
/* The film's pixel list: n_pixels x (x, y) uint32 into a host array. */
int rtmi_film_pixels(rtmi_film *film, uint32_t *xy);

/* Executor::run_* + tile_worker + pixel_worker (executor.rs:77-270) for the film's pixels: traces samples
 * [sample_begin, sample_end) of every pixel through the wavefront pipeline (camera ray, then per bounce: closest
 * hit, shade, compact) and adds them to the film (RaySeries::add_sample, renderer.rs:63-82).  Synchronous.
 * stats may be NULL. */
int rtmi_render(rtmi_scene *scene, rtmi_film *film, const rtmi_camera *camera, const rtmi_render_params *params,
                rtmi_render_stats *stats);

/* Tiler::tile_count / Tiler::tile (crates/rt-cli/src/tile.rs:69-97). out = x_start, x_end, y_start, y_end. */
uint32_t rtmi_tile_count(uint32_t width, uint32_t height, uint32_t grain);
int rtmi_tile_get(uint32_t offset_x, uint32_t offset_y, uint32_t width, uint32_t height, uint32_t grain,
                  uint32_t idx, uint32_t out[4]);

/* ---- several GPUs.  The reference shares one frame between the threads of one process (dispatch_async,
 * executor.rs:307-330); its unit of work sharing is the 32x32 tile (tile.rs:56-97).  Here every GPU (one rank each: a
 * process or a thread with its own rtmi_device) holds the whole scene, renders all samples of every world-th tile
 * into a film of its own (rtmi_film_create_tiles) and the per-pixel accumulators are then gathered into the root's
 * frame.  Samples depend only on (seed, x, y, sample), so the gathered frame is bit-identical to a one-GPU render. */

/* Rank `rank` of `world` owns tiles *out_first, *out_first + *out_stride, ...: *out_count of the n_tiles.  Interleaved,
 * because cost per tile varies by an order of magnitude with what the tile sees. */
int rtmi_tile_partition(uint32_t n_tiles, uint32_t world, uint32_t rank, uint32_t *out_first, uint32_t *out_stride,
                        uint32_t *out_count);

/* A communicator over RCCL (xGMI between the GPUs of a node).  One rank calls rtmi_comm_unique_id and hands the id to
 * the others by any means (a file, a socket, MPI, torch.distributed); then every rank calls rtmi_comm_create with it,
 * all at the same time (the call returns when all `world` ranks have joined).  RCCL is loaded when first needed; a
 * host without it gets RTMI_ERR_DEVICE here and can still use everything else. */
#define RTMI_COMM_ID_BYTES 128
int rtmi_comm_unique_id(unsigned char id[RTMI_COMM_ID_BYTES]);
int rtmi_comm_create(rtmi_device *dev, const unsigned char id[RTMI_COMM_ID_BYTES], int rank, int world,
                     rtmi_comm **out);
void rtmi_comm_destroy(rtmi_comm *comm);

/* The exchange step of a frame: copies every rank's film accumulators (RaySeries of its pixels) into `frame` on rank
 * `root`, at the pixels' positions there.  Collective: every rank of `comm` calls it with its own film; `frame` is
 * used on the root only (a film made by rtmi_film_create_frame on the root's device; NULL elsewhere) and pixels no
 * rank sent keep their old values.  Nothing is reduced: ranks must own disjoint pixels.  comm == NULL is the
 * single-GPU case (film and frame on one device).  The first call for a (film, comm, root) combination also sends
 * the pixel lists to the root; later calls move only the 88 bytes per pixel, each peer straight to the root (the root's
 * own pixels go from its film into the frame directly), after a 32-byte status exchange in which the ranks tell each
 * other which film they bring and whether they can take part.  Any rank may bring a different film from call to call:
 * the lists travel again whenever the set of films differs from the one a rank's plan was made for.  If a rank cannot
 * take part (a bad argument, a film with pixels outside the frame, no memory for the root's staging), every rank
 * returns an error (the failing rank its own, the others RTMI_ERR_STATE) and nothing has been exchanged.  The root may
 * gather into a different frame from call to call.  Synchronous: returns when the frame is complete on the root. */
int rtmi_film_gather(rtmi_film *film, rtmi_comm *comm, int root, rtmi_film *frame);
/* For a rank that cannot call rtmi_film_gather this frame (its render failed, it has no film): takes the rank's place
 * in the collective and tells the peers, which then return RTMI_ERR_STATE from their rtmi_film_gather instead of
 * waiting for this rank.  The reference has no counterpart (a panicking rayon worker aborts the whole process,
 * executor.rs:313-329). */
int rtmi_comm_gather_abort(rtmi_comm *comm);

#ifdef __cplusplus
}
#endif
#endif
