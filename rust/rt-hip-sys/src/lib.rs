//! FFI binding of librtmi.so (include/rtmi.h) and its mapping onto the reference's traits.
//!
//! NOT COMPILED in the build image (no Rust toolchain there); shipped as the source a maintainer of
//! philae-ael/Raytracing would add.  `HipScene` implements `rt::scene::SceneT` where `EmbreeScene` does
//! (crates/rt/src/aggregate/embree.rs:164-266), `CommittedHipScene` implements `rt::shape::Shape`
//! (embree.rs:100-162), and `run_hip` stands where `Executor::run_multithreaded` does
//! (crates/rt-cli/src/executor.rs:77-146).
use std::ffi::{c_char, c_int, c_void, CStr};

use rt::{
    material::{DiffuseBxDF, LightDescriptor, MaterialDescriptor, MaterialId},
    math::{bounds::Bounds, point::Point},
    ray::Ray,
    scene::SceneT,
    shape::{local_info, FullIntersectionResult, MinIntersectionResult, RayIntersection, Shape},
};

#[repr(C)]
pub struct RtmiMaterial { pub kind: u32, pub rgb: [f32; 3], pub ior: f32, pub roughness: f32, pub pad: [u32; 2] }
#[repr(C)]
pub struct RtmiRay { pub org: [f32; 3], pub tnear: f32, pub dir: [f32; 3], pub tfar: f32 }
#[repr(C)]
#[derive(Default)]
pub struct RtmiHit { pub t: f32, pub pos: [f32; 3], pub n: [f32; 3], pub uv: [f32; 2], pub geom_id: u32, pub prim_id: u32, pub mat_id: u32 }
#[repr(C)]
pub struct RtmiCamera { pub width: u32, pub height: u32, pub half_width: f32, pub half_height: f32, pub focal_length: f32, pub aperture: f32, pub lens: [f32; 3], pub rot: [f32; 4] }
#[repr(C)]
pub struct RtmiTransform { pub translation: [f32; 3], pub scale: [f32; 3], pub rot: [f32; 4] }
#[repr(C)]
pub struct RtmiRenderParams { pub spp: u32, pub sample_begin: u32, pub sample_end: u32, pub max_depth: u32, pub integrator: u32, pub samples_per_pass: u32, pub seed: u64, pub allowed_error: f32, pub flags: u32 }
#[repr(C)]
#[derive(Default)]
pub struct RtmiRenderStats { pub rays: u64, pub paths: u64, pub passes: u32, pub max_bounces: u32, pub gpu_ms: f32, pub bounce_ms: f32, pub bounce_launches: u32, pub extend_ms: f32, pub tail_rays: u64 }

pub enum RtmiDevice {}
pub enum RtmiScene {}
pub enum RtmiFilm {}
pub enum RtmiComm {}
pub const RTMI_COMM_ID_BYTES: usize = 128;

extern "C" {
    pub fn rtmi_last_error() -> *const c_char;
    pub fn rtmi_device_create(hip_ordinal: c_int, out: *mut *mut RtmiDevice) -> c_int;
    pub fn rtmi_device_destroy(dev: *mut RtmiDevice);
    pub fn rtmi_scene_create(dev: *mut RtmiDevice, out: *mut *mut RtmiScene) -> c_int;
    pub fn rtmi_scene_destroy(scene: *mut RtmiScene);
    pub fn rtmi_scene_insert_material(scene: *mut RtmiScene, mat: *const RtmiMaterial, out_id: *mut u32) -> c_int;
    pub fn rtmi_scene_insert_light(scene: *mut RtmiScene, pos: *const f32) -> c_int;
    pub fn rtmi_scene_insert_mesh(scene: *mut RtmiScene, material: u32, vertices: *const f32, n_vertices: usize, indices: *const u32, n_triangles: usize, out_geom: *mut u32) -> c_int;
    pub fn rtmi_scene_insert_sphere(scene: *mut RtmiScene, material: u32, center: *const f32, radius: f32, out_geom: *mut u32) -> c_int;
    pub fn rtmi_scene_load_obj(scene: *mut RtmiScene, path: *const c_char, transform: *const RtmiTransform, default_material: u32, out_first: *mut u32, out_n: *mut u32) -> c_int;
    pub fn rtmi_scene_commit(scene: *mut RtmiScene, progress: Option<extern "C" fn(f64, *mut c_void) -> c_int>, user: *mut c_void) -> c_int;
    pub fn rtmi_scene_bounds(scene: *const RtmiScene, lo: *mut f32, hi: *mut f32) -> c_int;
    pub fn rtmi_intersect(scene: *mut RtmiScene, rays: *const RtmiRay, hits: *mut RtmiHit, n: usize) -> c_int;
    pub fn rtmi_film_create(dev: *mut RtmiDevice, xy: *const u32, n_pixels: usize, d_accum: *mut f32, out: *mut *mut RtmiFilm) -> c_int;
    pub fn rtmi_film_destroy(film: *mut RtmiFilm);
    pub fn rtmi_film_resolve(film: *mut RtmiFilm, out: *mut f32, srgb: c_int) -> c_int;
    pub fn rtmi_intersect_device(scene: *mut RtmiScene, d_rays: *const RtmiRay, d_hits: *mut RtmiHit, n: usize, hip_stream: *mut c_void) -> c_int;
    pub fn rtmi_intersect_device_static(scene: *mut RtmiScene, d_rays: *const RtmiRay, d_hits: *mut RtmiHit, n: usize, hip_stream: *mut c_void) -> c_int;
    pub fn rtmi_film_create_tiles(dev: *mut RtmiDevice, offset_x: u32, offset_y: u32, width: u32, height: u32, grain: u32, first_tile: u32, tile_stride: u32, block_w: u32, block_h: u32, out: *mut *mut RtmiFilm) -> c_int;
    pub fn rtmi_film_create_frame(dev: *mut RtmiDevice, offset_x: u32, offset_y: u32, width: u32, height: u32, out: *mut *mut RtmiFilm) -> c_int;
    pub fn rtmi_film_pixel_count(film: *const RtmiFilm) -> usize;
    pub fn rtmi_film_pixels(film: *mut RtmiFilm, xy: *mut u32) -> c_int;
    pub fn rtmi_tile_partition(n_tiles: u32, world: u32, rank: u32, out_first: *mut u32, out_stride: *mut u32, out_count: *mut u32) -> c_int;
    pub fn rtmi_comm_unique_id(id: *mut u8) -> c_int;
    pub fn rtmi_comm_create(dev: *mut RtmiDevice, id: *const u8, rank: c_int, world: c_int, out: *mut *mut RtmiComm) -> c_int;
    pub fn rtmi_comm_destroy(comm: *mut RtmiComm);
    pub fn rtmi_film_gather(film: *mut RtmiFilm, comm: *mut RtmiComm, root: c_int, frame: *mut RtmiFilm) -> c_int;
    pub fn rtmi_comm_gather_abort(comm: *mut RtmiComm) -> c_int;
    pub fn rtmi_render(scene: *mut RtmiScene, film: *mut RtmiFilm, camera: *const RtmiCamera, params: *const RtmiRenderParams, stats: *mut RtmiRenderStats) -> c_int;
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(rtmi_last_error()).to_string_lossy().into_owned() }
}

/// The reference panics when geometry cannot be created (embree.rs:190-248); so does this.
fn must(rc: c_int, what: &str) {
    if rc != 0 { panic!("{what}: {}", last_error()); }
}

pub struct HipDevice(pub *mut RtmiDevice);
impl HipDevice {
    pub fn try_new(ordinal: Option<i32>) -> anyhow::Result<Self> {
        let mut d = std::ptr::null_mut();
        let rc = unsafe { rtmi_device_create(ordinal.unwrap_or(-1), &mut d) };
        if rc != 0 { anyhow::bail!("rtmi_device_create: {}", last_error()); }
        Ok(Self(d))
    }
}
impl Drop for HipDevice { fn drop(&mut self) { unsafe { rtmi_device_destroy(self.0) } } }

pub struct HipScene<'a> {
    pub device: &'a HipDevice,
    raw: *mut RtmiScene,
    pub materials: Vec<MaterialDescriptor>,
    pub lights: Vec<Point>,
}

/// A BxDF has to say what it is to cross the boundary; upstream adds `fn as_pod(&self) -> RtmiMaterial` to the
/// `BxDF` trait (crates/rt/src/material/mod.rs:40-54), implemented by the four BxDFs (kinds 0..3).
pub trait AsRtmiMaterial { fn as_pod(&self) -> RtmiMaterial; }

impl<'a> HipScene<'a> {
    pub fn new(device: &'a HipDevice) -> Self {
        let mut raw = std::ptr::null_mut();
        must(unsafe { rtmi_scene_create(device.0, &mut raw) }, "rtmi_scene_create");
        // rtmi_scene_create has already made material 0 the black "Sky" diffuse; keep the host-side table in step
        // (embree.rs:41-46), so that `world.materials[id.0]` finds what the C side calls `id`.
        let sky = MaterialDescriptor { label: Some("Sky".into()), material: Box::new(DiffuseBxDF { albedo: [0.0, 0.0, 0.0].into() }) };
        Self { device, raw, materials: vec![sky], lights: Vec::new() }
    }
    /// World { objects, lights, materials, world_material } over this scene (embree.rs:90-97).
    pub fn sky_material(&self) -> MaterialId { MaterialId(0) }
    pub fn commit(&mut self) -> anyhow::Result<CommittedHipScene<'_, 'a>> {
        let rc = unsafe { rtmi_scene_commit(self.raw, None, std::ptr::null_mut()) };
        if rc != 0 { anyhow::bail!("rtmi_scene_commit: {}", last_error()); }
        Ok(CommittedHipScene { scene: self })
    }
    pub fn commit_with_progress<F: FnMut(f64) -> bool>(&mut self, mut f: F) -> anyhow::Result<CommittedHipScene<'_, 'a>> {
        extern "C" fn tramp<F: FnMut(f64) -> bool>(amount: f64, user: *mut c_void) -> c_int {
            let f = unsafe { &mut *(user as *mut F) };
            f(amount) as c_int
        }
        let rc = unsafe { rtmi_scene_commit(self.raw, Some(tramp::<F>), &mut f as *mut F as *mut c_void) };
        if rc != 0 { anyhow::bail!("rtmi_scene_commit: {}", last_error()); }
        Ok(CommittedHipScene { scene: self })
    }
}
impl Drop for HipScene<'_> { fn drop(&mut self) { unsafe { rtmi_scene_destroy(self.raw) } } }

impl SceneT for HipScene<'_> {
    type GeometryHandle = u32;
    fn insert_material(&mut self, mat: MaterialDescriptor) -> MaterialId {
        let pod = mat.material.as_pod();
        let mut id = 0u32;
        must(unsafe { rtmi_scene_insert_material(self.raw, &pod, &mut id) }, "insert_material");
        self.materials.push(mat);
        debug_assert_eq!(id as usize, self.materials.len() - 1, "host and device material tables out of step");
        MaterialId(id as usize)
    }
    fn insert_light(&mut self, light: LightDescriptor) {
        let p = light.light_pos.vec().to_array();
        must(unsafe { rtmi_scene_insert_light(self.raw, p.as_ptr()) }, "insert_light");
        self.lights.push(light.light_pos);
    }
    fn insert_mesh(&mut self, material: MaterialId, vertices: &[[f32; 3]], indices: &[[u32; 3]]) -> u32 {
        let mut g = 0u32;
        must(unsafe { rtmi_scene_insert_mesh(self.raw, material.0 as u32, vertices.as_ptr() as *const f32, vertices.len(), indices.as_ptr() as *const u32, indices.len(), &mut g) }, "insert_mesh");
        g
    }
    fn insert_sphere(&mut self, material: MaterialId, origin: Point, radius: f32) -> u32 {
        let c = origin.vec().to_array();
        let mut g = 0u32;
        must(unsafe { rtmi_scene_insert_sphere(self.raw, material.0 as u32, c.as_ptr(), radius, &mut g) }, "insert_sphere");
        g
    }
}

pub struct CommittedHipScene<'c, 'a> { scene: &'c HipScene<'a> }
unsafe impl Send for CommittedHipScene<'_, '_> {}
unsafe impl Sync for CommittedHipScene<'_, '_> {}

impl CommittedHipScene<'_, '_> {
    /// embree.rs:90-97
    pub fn into_world(&self) -> anyhow::Result<rt::renderer::World> {
        Ok(rt::renderer::World { objects: self, lights: &self.scene.lights, materials: &self.scene.materials, world_material: self.scene.sky_material() })
    }
}

impl Shape for CommittedHipScene<'_, '_> {
    /// One ray per call keeps the trait working; integrators that stay on the CPU should batch through
    /// `rtmi_intersect` instead, and the GPU executor below does not call this at all.
    fn intersection_full(&self, ray: Ray) -> FullIntersectionResult {
        let r = RtmiRay { org: ray.origin.vec().to_array(), tnear: ray.bounds.0, dir: ray.direction.to_array(), tfar: ray.bounds.1 };
        let mut h = RtmiHit::default();
        must(unsafe { rtmi_intersect(self.scene.raw, &r, &mut h, 1) }, "rtmi_intersect");
        if h.geom_id == u32::MAX { return FullIntersectionResult::NoIntersection; }
        FullIntersectionResult::Intersection(RayIntersection {
            t: h.t,
            local_info: local_info::Full { pos: Point::new(h.pos[0], h.pos[1], h.pos[2]), normal: h.n.into(), material: MaterialId(h.mat_id as usize), uv: h.uv },
        })
    }
    fn intersect_bare(&self, _ray: Ray) -> MinIntersectionResult { todo!() } // as upstream, embree.rs:142-144
    fn bounding_box(&self) -> Bounds {
        let (mut lo, mut hi) = ([0f32; 3], [0f32; 3]);
        must(unsafe { rtmi_scene_bounds(self.scene.raw, lo.as_mut_ptr(), hi.as_mut_ptr()) }, "rtmi_scene_bounds");
        Bounds { origin: Point::new(lo[0], lo[1], lo[2]), end: Point::new(hi[0], hi[1], hi[2]) }
    }
}

/// GPU replacement of `Executor::run_multithreaded`: renders `samples` of the listed pixels and hands back the seven
/// channels per pixel (15 floats, order of `RaySeries::as_pixelresult`, crates/rt/src/renderer.rs:51-59), which the
/// caller slices per tile into `TileMsg`s.
pub fn run_hip(scene: &CommittedHipScene, camera: &RtmiCamera, params: &RtmiRenderParams, xy: &[[u32; 2]]) -> anyhow::Result<(Vec<f32>, RtmiRenderStats)> {
    let mut film = std::ptr::null_mut();
    let rc = unsafe { rtmi_film_create(scene.scene.device.0, xy.as_ptr() as *const u32, xy.len(), std::ptr::null_mut(), &mut film) };
    if rc != 0 { anyhow::bail!("rtmi_film_create: {}", last_error()); }
    let mut stats = RtmiRenderStats::default();
    let mut out = vec![0f32; 15 * xy.len()];
    let rc = unsafe { rtmi_render(scene.scene.raw, film, camera, params, &mut stats) };
    let rc2 = if rc == 0 { unsafe { rtmi_film_resolve(film, out.as_mut_ptr(), 0) } } else { rc };
    unsafe { rtmi_film_destroy(film) };
    if rc2 != 0 { anyhow::bail!("rtmi_render: {}", last_error()); }
    Ok((out, stats))
}

/// One rank of a multi-GPU render (one thread or process per GPU, each with its own `HipDevice` and a full copy of the
/// scene): renders this rank's interleaved share of the 32x32 tiles and gathers every rank's accumulators into rank 0's
/// frame.  `id` comes from `rtmi_comm_unique_id` on one rank and reaches the others by any channel.  Returns the 15
/// resolved channel planes of the whole frame on rank 0, an empty Vec elsewhere.
pub fn run_hip_rank(scene: &CommittedHipScene, camera: &RtmiCamera, params: &RtmiRenderParams, tile_size: u32,
                    id: &[u8; RTMI_COMM_ID_BYTES], rank: i32, world: i32) -> anyhow::Result<Vec<f32>> {
    let dev = scene.scene.device.0;
    let (w, h) = (camera.width, camera.height);
    let fail = |what: &str| anyhow::anyhow!("{what}: {}", last_error());
    unsafe {
        // Without a communicator there is nothing to leave in step with; every later failure must still take this
        // rank's turn in the gather (rtmi_comm_gather_abort), or the peers would wait for it.
        let mut comm = std::ptr::null_mut();
        if world > 1 && rtmi_comm_create(dev, id.as_ptr(), rank, world, &mut comm) != 0 { return Err(fail("rtmi_comm_create")); }
        let mut film = std::ptr::null_mut();
        let mut frame = std::ptr::null_mut();
        let mut err = None;
        if rtmi_film_create_tiles(dev, 0, 0, w, h, tile_size, rank as u32, world as u32, 8, 8, &mut film) != 0 {
            err = Some(fail("rtmi_film_create_tiles"));
        } else if rank == 0 && rtmi_film_create_frame(dev, 0, 0, w, h, &mut frame) != 0 {
            err = Some(fail("rtmi_film_create_frame"));
        } else {
            let mut stats = RtmiRenderStats::default();
            if rtmi_render(scene.scene.raw, film, camera, params, &mut stats) != 0 { err = Some(fail("rtmi_render")); }
        }
        let mut out = Vec::new();
        if err.is_some() {
            if !comm.is_null() { rtmi_comm_gather_abort(comm); } // the peers' rtmi_film_gather returns an error
        } else if rtmi_film_gather(film, comm, 0, frame) != 0 {
            err = Some(fail("rtmi_film_gather"));
        } else if rank == 0 {
            out = vec![0f32; 15 * (w as usize) * (h as usize)];
            if rtmi_film_resolve(frame, out.as_mut_ptr(), 0) != 0 { err = Some(fail("rtmi_film_resolve")); }
        }
        if !film.is_null() { rtmi_film_destroy(film); }
        if !frame.is_null() { rtmi_film_destroy(frame); }
        if !comm.is_null() { rtmi_comm_destroy(comm); }
        match err { Some(e) => Err(e), None => Ok(out) }
    }
}
