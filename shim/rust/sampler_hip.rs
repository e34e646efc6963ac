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
// Environment knobs read by the library itself (nothing to do here): MRT_GPUS=N (row-shard over N devices, one RCCL
// gather per execute), MRT_DEFER=1 (per-sample execute calls only book their sample; the frame is traced in batches
// when img() observes it -- the per-sample Duration logged at src/cli.rs:164 is then ~0).
use std::ffi::CStr;
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
    fn mrt_last_error() -> *const c_char;
}

fn last_error() -> String { unsafe { CStr::from_ptr(mrt_last_error()).to_string_lossy().into_owned() } }

pub struct Sampler { ctx: *mut MrtCtx, seed: u64 }
unsafe impl Send for Sampler {}          // one owner at a time, like &mut self (HttpServer: one Sampler per thread)

impl Sampler {
    pub fn new(_workers: u32, _n_dim: usize) -> Sampler {
        let seed = std::env::var("MRT_SEED").ok().and_then(|s| s.parse().ok())
            .unwrap_or_else(|| rand::random::<u64>());          // the reference is unseeded: default stays random
        Sampler { ctx: std::ptr::null_mut(), seed }
    }

    pub fn execute<'a>(&mut self, scene: &'a Scene, frame: &Frame, rt: &'a RayTracer) -> Duration {
        if self.ctx.is_null() {
            self.ctx = create(scene, frame, rt, self.seed).unwrap_or_else(|e| panic!("{e}"));   // reference panics mid-render
        }
        let mut secs = 0f64;
        if unsafe { mrt_execute(self.ctx, 1, &mut secs) } != 0 { panic!("{}", last_error()); }
        Duration::from_secs_f64(secs)
    }

    pub fn img(&self, frame: &Frame) -> Result<RgbImage, String> {
        if self.ctx.is_null() { return Err("img before execute".into()); }
        let mut buf = vec![0u8; frame.res.0 as usize * frame.res.1 as usize * 3];
        if unsafe { mrt_img(self.ctx, buf.as_mut_ptr()) } != 0 { return Err(last_error()); }
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
    // flags: MRT_FLAG_NO_EVENT_TIMING (2) -- this caller runs one sample per call and never reads mrt_stats
    let opts = MrtOpts { abi_version: 2, seed, device: -1, shard_index: 0, shard_count: 1, shard_rows: 0, n_devices: 0, flags: 2, reserved: [0; 4] };
    let ctx = unsafe { mrt_create(&desc, &opts) };
    if ctx.is_null() { Err(last_error()) } else { Ok(ctx) }
}
