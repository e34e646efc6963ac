// sampler_hip.rs -- the body of micro-raytracer's src/sampler.rs under `--features hip`.
//
// Replaces src/sampler.rs:11-100 (the scoped_threadpool / HashMap Sampler) by calls into the C ABI of
// include/mrt.h (libmrt_hip.so).  Same three signatures, so CLI::raytrace (src/cli.rs:155-177) and
// HttpServer::raytrace (src/http.rs:136-148) compile unchanged:
//     #[cfg(feature = "hip")]      mod sampler { include!("sampler_hip.rs"); }
//     #[cfg(not(feature = "hip"))] ... the existing CPU implementation ...
// No Rust toolchain exists in the image this repository is built in: the file is delivered as source.  What can be
// checked without rustc is checked: tests/test_shim_layout.py parses the #[repr(C)] declarations below, lays them out
// by the C rules repr(C) prescribes and compares every size and field offset with what the C compiler reports for
// include/mrt.h (tests/native/layout.c) and with the committed table shim/rust/layout.txt.
//
// Per-sample callers: CLI::raytrace and HttpServer::raytrace call execute once per sample (src/cli.rs:162-170,
// src/http.rs:141-144).  The shim therefore creates its context with MRT_FLAG_DEFER: an execute call only books its
// sample (the Duration it returns, logged at src/cli.rs:164, is ~0) and the frame is traced in 1024-sample batches
// when img() observes it -- the batched kernel (7.0 instead of 5.4 Gsamples/s at 1080p) for the unmodified binary; with
// --update (img() after every sample) this degenerates to the eager loop by itself.  MRT_DEFER=0 in the environment
// keeps every call synchronous (real per-sample Durations).  Other knobs, read by the library itself: MRT_GPUS=N
// (row-shard over N devices, one RCCL gather per execute), MRT_SEED (read here).
use std::collections::hash_map::DefaultHasher;
use std::ffi::CStr;
use std::hash::Hasher;
use std::os::raw::{c_char, c_int};
use std::time::Duration;
use image::RgbImage;
use crate::rt::{RayTracer, Scene, Frame, RendererKind, LightKind, Texture};

#[repr(C)] struct MrtCamera { pos: [f32; 3], dir: [f32; 4], fov: f32, gamma: f32, exp: f32, aprt: f32, foc: f32 }
#[repr(C)] struct MrtFrame { res_w: u16, res_h: u16, ssaa: f32, cam: MrtCamera }
#[repr(C)] struct MrtRt { bounce: u32, sample: u32, loss: f32 }
#[repr(C)] struct MrtTexture { w: u32, h: u32, dat: *const f32 }
#[repr(C)] #[derive(Clone, Copy)]
struct MrtMaterial { albedo: [f32; 3], rough: f32, metal: f32, glass: f32, opacity: f32, emit: f32,
                     tex: i32, rmap: i32, mmap: i32, gmap: i32, omap: i32, emap: i32 }
#[repr(C)] struct MrtInstance { pos: [f32; 3], dir: [f32; 4] }
#[repr(C)] struct MrtRenderer { kind: u32, param: [f32; 9], tris: *const f32, n_tris: u32, mat: MrtMaterial,
                                inst: *const MrtInstance, n_inst: u32 }
#[repr(C)] struct MrtLight { kind: u32, v: [f32; 3], pwr: f32, color: [f32; 3] }
#[repr(C)] struct MrtSky { color: [f32; 3], pwr: f32 }
#[repr(C)] struct MrtScene { renderer: *const MrtRenderer, n_renderer: u32, light: *const MrtLight, n_light: u32,
                             sky: MrtSky, textures: *const MrtTexture, n_textures: u32 }
#[repr(C)] struct MrtRenderDesc { rt: MrtRt, frame: MrtFrame, scene: MrtScene }
#[repr(C)] struct MrtOpts { abi_version: u32, seed: u64, device: i32, shard_index: u32, shard_count: u32,
                            shard_rows: u32, n_devices: u32, flags: u32, reserved: [u32; 4] }
#[repr(C)] struct MrtCtx { _private: [u8; 0] }

extern "C" {
    fn mrt_create(desc: *const MrtRenderDesc, opts: *const MrtOpts) -> *mut MrtCtx;
    fn mrt_destroy(ctx: *mut MrtCtx);
    fn mrt_execute(ctx: *mut MrtCtx, n_samples: u32, seconds: *mut f64) -> c_int;
    fn mrt_img(ctx: *mut MrtCtx, rgb8: *mut u8) -> c_int;
    fn mrt_dims(ctx: *const MrtCtx, nw: *mut u32, nh: *mut u32, local_rows: *mut u32) -> c_int;
    fn mrt_accum(ctx: *mut MrtCtx, rgb: *mut f32, count: *mut u32) -> c_int;
    fn mrt_set_accum(ctx: *mut MrtCtx, rgb: *const f32, count: u32) -> c_int;
    fn mrt_last_error() -> *const c_char;
}

fn last_error() -> String { unsafe { CStr::from_ptr(mrt_last_error()).to_string_lossy().into_owned() } }

pub struct Sampler { ctx: *mut MrtCtx, seed: u64, print: u64, stale: bool }
unsafe impl Send for Sampler {}          // one owner at a time, like &mut self (HttpServer: one Sampler per thread)

/// What the device context was built from: execute() receives scene, frame and rt on EVERY call (src/sampler.rs:28), so
/// a caller may pass something else the next time.  Scalars and instance lists by value, bulk data (mesh triangles,
/// texels) by address, length and a strided sample -- cheap enough to recompute per call.
fn fingerprint(scene: &Scene, frame: &Frame, rt: &RayTracer) -> u64 {
    let mut h = DefaultHasher::new();
    let mut f = |v: f32| h.write_u32(v.to_bits());
    let c = &frame.cam;
    for v in [frame.ssaa, c.pos.x, c.pos.y, c.pos.z, c.dir.w, c.dir.x, c.dir.y, c.dir.z, c.fov, c.gamma, c.exp, c.aprt, c.foc,
              rt.loss, scene.sky.color.x, scene.sky.color.y, scene.sky.color.z, scene.sky.pwr] { f(v); }
    let tex = |t: &Option<Texture>, h: &mut DefaultHasher| if let Some(t) = t {
        h.write_usize(t.w); h.write_usize(t.h);
        if let Some(d) = &t.dat { h.write_usize(d.as_ptr() as usize); h.write_usize(d.len());
            for v in d.iter().step_by(1 + d.len() / 256) { h.write_u32(v.x.to_bits()); h.write_u32(v.y.to_bits()); h.write_u32(v.z.to_bits()); } }
    } else { h.write_u8(0) };
    drop(f);
    h.write_u16(frame.res.0); h.write_u16(frame.res.1); h.write_usize(rt.bounce);
    for obj in scene.renderer.as_deref().unwrap_or(&[]) {
        let m = &obj.mat;
        for v in [m.albedo.x, m.albedo.y, m.albedo.z, m.rough, m.metal, m.glass, m.opacity, m.emit] { h.write_u32(v.to_bits()); }
        for t in [&m.tex, &m.rmap, &m.mmap, &m.gmap, &m.omap, &m.emap] { tex(t, &mut h); }
        match &obj.kind {
            RendererKind::Sphere(s) => { h.write_u8(0); h.write_u32(s.0.to_bits()); }
            RendererKind::Plane(p) => { h.write_u8(1); for v in [p.0.x, p.0.y, p.0.z] { h.write_u32(v.to_bits()); } }
            RendererKind::Box(b) => { h.write_u8(2); for v in [b.0.x, b.0.y, b.0.z] { h.write_u32(v.to_bits()); } }
            RendererKind::Triangle(t) => { h.write_u8(3); for v in [t.0.x, t.0.y, t.0.z, t.1.x, t.1.y, t.1.z, t.2.x, t.2.y, t.2.z] { h.write_u32(v.to_bits()); } }
            RendererKind::Mesh(me) => { h.write_u8(4); h.write_usize(me.mesh.as_ptr() as usize); h.write_usize(me.mesh.len());
                for t in me.mesh.iter().step_by(1 + me.mesh.len() / 256) { for v in [t.0.x, t.0.y, t.0.z, t.1.x, t.2.x] { h.write_u32(v.to_bits()); } } }
        }
        for i in obj.instance.iter() { for v in [i.pos.x, i.pos.y, i.pos.z, i.dir.w, i.dir.x, i.dir.y, i.dir.z] { h.write_u32(v.to_bits()); } }
    }
    for l in scene.light.as_deref().unwrap_or(&[]) {
        let v = match l.kind { LightKind::Point { pos } => (0u8, pos), LightKind::Dir { dir } => (1u8, dir) };
        h.write_u8(v.0);
        for x in [v.1.x, v.1.y, v.1.z, l.pwr, l.color.x, l.color.y, l.color.z] { h.write_u32(x.to_bits()); }
    }
    h.finish()
}

impl Sampler {
    pub fn new(_workers: u32, _n_dim: usize) -> Sampler {
        let seed = std::env::var("MRT_SEED").ok().and_then(|s| s.parse().ok())
            .unwrap_or_else(|| rand::random::<u64>());          // the reference is unseeded: default stays random
        Sampler { ctx: std::ptr::null_mut(), seed, print: 0, stale: false }
    }

    pub fn execute<'a>(&mut self, scene: &'a Scene, frame: &Frame, rt: &'a RayTracer) -> Duration {
        let print = fingerprint(scene, frame, rt);
        if self.ctx.is_null() || print != self.print || self.stale {
            self.stale = false;
            // first call, or the caller passes another scene / frame / rt than last time: the context is (re)built from what
            // is passed NOW.  Like the reference, whose map keeps adding whatever the scene (src/sampler.rs:60-70), the sums
            // accumulated so far are carried over when the supersampled frame keeps its size.
            let fresh = create(scene, frame, rt, self.seed).unwrap_or_else(|e| panic!("{e}"));   // reference panics mid-render
            if !self.ctx.is_null() {
                let (mut ow, mut oh, mut nw, mut nh) = (0u32, 0u32, 0u32, 0u32);
                unsafe { mrt_dims(self.ctx, &mut ow, &mut oh, std::ptr::null_mut()); mrt_dims(fresh, &mut nw, &mut nh, std::ptr::null_mut()); }
                if (ow, oh) == (nw, nh) {
                    let mut sums = vec![0f32; nw as usize * nh as usize * 3];
                    let mut count = 0u32;
                    if unsafe { mrt_accum(self.ctx, sums.as_mut_ptr(), &mut count) } != 0 { panic!("{}", last_error()); }
                    if count > 0 && unsafe { mrt_set_accum(fresh, sums.as_ptr(), count) } != 0 { panic!("{}", last_error()); }
                }
                unsafe { mrt_destroy(self.ctx) };
            }
            self.ctx = fresh;
            self.print = print;
        }
        let mut secs = 0f64;
        if unsafe { mrt_execute(self.ctx, 1, &mut secs) } != 0 { panic!("{}", last_error()); }
        Duration::from_secs_f64(secs)
    }

    /// The fingerprint sees bulk data (mesh triangles, texels) by address, length and a strided sample: an in-place edit
    /// between the sample points goes unnoticed.  A caller that edits such data in place calls this; the next execute()
    /// rebuilds the context from what it is passed.  (The reference's own callers never edit a scene mid-render.)
    pub fn invalidate(&mut self) { self.stale = true; }

    pub fn img(&self, frame: &Frame) -> Result<RgbImage, String> {
        if self.ctx.is_null() { return Err("img before execute".into()); }
        let mut buf = vec![0u8; frame.res.0 as usize * frame.res.1 as usize * 3];
        if unsafe { mrt_img(self.ctx, buf.as_mut_ptr()) } != 0 { return Err(last_error()); }     // settles booked samples first
        RgbImage::from_raw(frame.res.0 as u32, frame.res.1 as u32, buf).ok_or("bad image size".to_string())
    }
}

impl Drop for Sampler { fn drop(&mut self) { if !self.ctx.is_null() { unsafe { mrt_destroy(self.ctx) } } } }

/// Flatten rt::Render (src/rt.rs:10-190) into the POD descriptor.  Renderer order and per-renderer instance
/// order are preserved (first-minimum tie rule of src/rt.rs:872).  Pointers are borrowed only during mrt_create.
fn create(scene: &Scene, frame: &Frame, rt: &RayTracer, seed: u64) -> Result<*mut MrtCtx, String> {
    let mut tex_flat: Vec<Vec<f32>> = Vec::new();
    let mut textures: Vec<MrtTexture> = Vec::new();
    let mut add_tex = |t: &Option<Texture>| -> i32 {
        match t {
            None => -1,
            Some(t) => {
                let flat: Vec<f32> = t.dat.as_ref().map(|d| d.iter().flat_map(|v| [v.x, v.y, v.z]).collect()).unwrap_or_default();
                tex_flat.push(flat);
                let p = if t.dat.is_some() { tex_flat.last().unwrap().as_ptr() } else { std::ptr::null() };
                textures.push(MrtTexture { w: t.w as u32, h: t.h as u32, dat: p });
                (textures.len() - 1) as i32
            }
        }
    };
    let mut tris_flat: Vec<Vec<f32>> = Vec::new();
    let mut insts: Vec<Vec<MrtInstance>> = Vec::new();
    let mut rends: Vec<MrtRenderer> = Vec::new();
    for obj in scene.renderer.as_deref().unwrap_or(&[]) {
        let mut param = [0f32; 9];
        let (kind, tris): (u32, Vec<f32>) = match &obj.kind {
            RendererKind::Sphere(s) => { param[0] = s.0; (0, vec![]) }
            RendererKind::Plane(p) => { param[..3].copy_from_slice(&[p.0.x, p.0.y, p.0.z]); (1, vec![]) }
            RendererKind::Box(b) => { param[..3].copy_from_slice(&[b.0.x, b.0.y, b.0.z]); (2, vec![]) }
            RendererKind::Triangle(t) => { param = [t.0.x, t.0.y, t.0.z, t.1.x, t.1.y, t.1.z, t.2.x, t.2.y, t.2.z]; (3, vec![]) }
            RendererKind::Mesh(m) => (4, m.mesh.iter().flat_map(|t| [t.0.x, t.0.y, t.0.z, t.1.x, t.1.y, t.1.z, t.2.x, t.2.y, t.2.z]).collect()),
        };
        tris_flat.push(tris);
        insts.push(obj.instance.iter().map(|i| MrtInstance { pos: [i.pos.x, i.pos.y, i.pos.z], dir: [i.dir.w, i.dir.x, i.dir.y, i.dir.z] }).collect());
        let m = &obj.mat;
        let mat = MrtMaterial { albedo: [m.albedo.x, m.albedo.y, m.albedo.z], rough: m.rough, metal: m.metal, glass: m.glass,
            opacity: m.opacity, emit: m.emit, tex: add_tex(&m.tex), rmap: add_tex(&m.rmap), mmap: add_tex(&m.mmap),
            gmap: add_tex(&m.gmap), omap: add_tex(&m.omap), emap: add_tex(&m.emap) };
        rends.push(MrtRenderer { kind, param, tris: tris_flat.last().unwrap().as_ptr(), n_tris: (tris_flat.last().unwrap().len() / 9) as u32,
            mat, inst: insts.last().unwrap().as_ptr(), n_inst: insts.last().unwrap().len() as u32 });
    }
    let lights: Vec<MrtLight> = scene.light.as_deref().unwrap_or(&[]).iter().map(|l| match l.kind {
        LightKind::Point { pos } => MrtLight { kind: 0, v: [pos.x, pos.y, pos.z], pwr: l.pwr, color: [l.color.x, l.color.y, l.color.z] },
        LightKind::Dir { dir } => MrtLight { kind: 1, v: [dir.x, dir.y, dir.z], pwr: l.pwr, color: [l.color.x, l.color.y, l.color.z] },
    }).collect();
    let c = &frame.cam;
    let desc = MrtRenderDesc {
        rt: MrtRt { bounce: rt.bounce as u32, sample: rt.sample as u32, loss: rt.loss },
        frame: MrtFrame { res_w: frame.res.0, res_h: frame.res.1, ssaa: frame.ssaa,
            cam: MrtCamera { pos: [c.pos.x, c.pos.y, c.pos.z], dir: [c.dir.w, c.dir.x, c.dir.y, c.dir.z], fov: c.fov, gamma: c.gamma, exp: c.exp, aprt: c.aprt, foc: c.foc } },
        scene: MrtScene { renderer: rends.as_ptr(), n_renderer: rends.len() as u32, light: lights.as_ptr(), n_light: lights.len() as u32,
            sky: MrtSky { color: [scene.sky.color.x, scene.sky.color.y, scene.sky.color.z], pwr: scene.sky.pwr },
            textures: textures.as_ptr(), n_textures: textures.len() as u32 },
    };
    // flags: MRT_FLAG_NO_EVENT_TIMING (2) -- this caller runs one sample per call and never reads mrt_stats -- and
    // MRT_FLAG_DEFER (4) unless MRT_DEFER=0: per-sample calls book their sample, img() traces them batched
    let defer = std::env::var("MRT_DEFER").map(|v| v.trim() != "0").unwrap_or(true);
    let opts = MrtOpts { abi_version: 3, seed, device: -1, shard_index: 0, shard_count: 1, shard_rows: 0, n_devices: 0,
                         flags: 2 | if defer { 4 } else { 0 }, reserved: [0; 4] };
    let ctx = unsafe { mrt_create(&desc, &opts) };
    if ctx.is_null() { Err(last_error()) } else { Ok(ctx) }
}
