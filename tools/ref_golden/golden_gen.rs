//! golden_gen.rs -- reference-side generator of the fixtures that PIN sdfgrid-mi355x's oracle.
//!
//! This file is not part of sdf-viewer and not a copy of any of it: it is new test code that CALLS the reference
//! (`SDFDemo::sample`, `SDFSurface::normal`, `three_d::Srgba::from(..).to_linear_srgb()`, `LoadingManager`,
//! `SDFViewer::update`) and writes what they return, bit for bit, as JSON.  It exists because the image sdfgrid-mi355x is
//! built in has no Rust toolchain: whoever has one runs this ONCE and commits the three files it writes.
//!
//! How to run (ten minutes, see tools/ref_golden/README.md):
//!   1. cp golden_gen.rs <sdf-viewer checkout>/src/app/scene/sdf/golden_gen.rs
//!   2. append to <sdf-viewer checkout>/src/app/scene/sdf/mod.rs:      #[cfg(test)] mod golden_gen;
//!   3. SDFV_GOLDEN_OUT=/tmp/ref_golden cargo test --lib golden_gen -- --nocapture --test-threads=1
//!   4. cp /tmp/ref_golden/ref_*.json <sdfgrid-mi355x>/tests/golden/  &&  python -m pytest tests -q -k ref_golden
//!
//! It is a child module of `app::scene::sdf` so that it may name that module's private items (`AIR_DIST`,
//! `SDFViewer::texture_from_data`); nothing here needs a GL context (see `viewer_without_gl`).
//!
//! Every f32 is written as the 8 hex digits of `f32::to_bits` -- no decimal round trip.

use std::fmt::Write as _;
use std::mem::MaybeUninit;
use std::ptr::addr_of_mut;

use cgmath::Vector3;
use clap::Parser;
use three_d::{Srgba, TextureData};

use super::loading::LoadingManager;
use super::{SDFViewer, AIR_DIST};
use crate::sdf::demo::SDFDemo;
use crate::sdf::{SDFParamValue, SDFSurface};

fn out_dir() -> std::path::PathBuf {
    let d = std::env::var("SDFV_GOLDEN_OUT").unwrap_or_else(|_| "target/ref_golden".to_string());
    std::fs::create_dir_all(&d).expect("create output directory");
    std::path::PathBuf::from(d)
}

fn hx(v: f32) -> String {
    format!("\"{:08x}\"", v.to_bits())
}

fn hx_list(vs: &[f32]) -> String {
    let mut s = String::from("[");
    for (i, v) in vs.iter().enumerate() {
        if i > 0 {
            s.push(',');
        }
        s.push_str(&hx(*v));
    }
    s.push(']');
    s
}

/// The demo configurations sampled: clap flags exactly as a user would pass them (cube.rs:15-18, sphere.rs:11-14,
/// demo/mod.rs:26-29), with the parameter values they mean spelled out for the consumer.
struct Config {
    flags: &'static [&'static str],
    /// cube_half_side, cube_material (0 brick / 1 normal), sphere_radius, sphere_material, max_distance_custom_material,
    /// disable_sphere
    params: (f32, u32, f32, u32, f32, u32),
    seeded_points: usize,
}

const CONFIGS: &[Config] = &[
    Config { flags: &[], params: (0.95, 0, 1.05, 1, 0.05, 0), seeded_points: 4096 },
    Config {
        flags: &["--cube-material", "normal", "--sphere-material", "brick"],
        params: (0.95, 1, 1.05, 0, 0.05, 0),
        seeded_points: 1024,
    },
    Config { flags: &["--disable-sphere", "true"], params: (0.95, 0, 1.05, 1, 0.05, 1), seeded_points: 512 },
    Config {
        flags: &["--cube-half-side", "0.8", "--sphere-radius", "0.3", "--max-distance-custom-material", "0.25"],
        params: (0.8, 0, 0.3, 1, 0.25, 0),
        seeded_points: 512,
    },
];

fn demo_from_flags(flags: &[&str]) -> SDFDemo {
    let mut argv: Vec<&str> = vec!["demo"];
    argv.extend_from_slice(flags);
    SDFDemo::parse_from(argv)
}

/// SURVEY's known-answer points, the voxel-coordinate asymmetry around the centre of a 64-grid, points ON the faces
/// and corners, and `n` points of a xorshift32 stream over [-1.25, 1.25)^3 (the points travel in the file: the consumer
/// does not have to regenerate them).
fn points(n: usize) -> Vec<[f32; 3]> {
    let mut p: Vec<[f32; 3]> = vec![
        [0.0, 0.0, 0.0],
        [1.0, 1.0, 1.0],
        [0.9, 0.6, 0.0],
        [0.96, 0.3, -0.2],
        [-1.0, 0.968_253_97, 0.015_873_075],
        [-0.015_873_015, 0.015_873_075, 1.0],
        [0.95, 0.95, 0.95],
        [-0.95, 0.2, 0.95],
        [0.0, 1.05, 0.0],
        [0.0, -1.0, 0.0],
        [-0.0, 0.0, -0.0],
        [0.97, -0.97, 0.5],
    ];
    let mut s: u32 = 0x9E37_79B9;
    let mut next = move || {
        s ^= s << 13;
        s ^= s >> 17;
        s ^= s << 5;
        ((s >> 8) as f32) * (1.0 / 16_777_216.0) * 2.5 - 1.25
    };
    for _ in 0..n {
        let x = next();
        let y = next();
        let z = next();
        p.push([x, y, z]);
    }
    p
}

fn sample_words(sdf: &dyn SDFSurface, p: [f32; 3], distance_only: bool) -> String {
    let s = sdf.sample(Vector3::new(p[0], p[1], p[2]), distance_only);
    hx_list(&[s.distance, s.color.x, s.color.y, s.color.z, s.metallic, s.roughness, s.occlusion])
}

fn normal_words(sdf: &dyn SDFSurface, p: [f32; 3], eps: Option<f32>) -> String {
    let n = sdf.normal(Vector3::new(p[0], p[1], p[2]), eps);
    hx_list(&[n.x, n.y, n.z])
}

/// ref_samples.json: SDFSurface::sample / ::normal of the demo tree (root id 0, children()[0] = cube id 1, children()[1]
/// = sphere id 2 -- demo/mod.rs:78-81) at every point, for every configuration.
#[test]
fn golden_gen_samples() {
    let mut o = String::new();
    write!(o, "{{\"generator\":\"sdf-viewer golden_gen v1\",\"crate_version\":\"{}\",\"configs\":[", env!("CARGO_PKG_VERSION")).unwrap();
    for (ci, cfg) in CONFIGS.iter().enumerate() {
        let demo = demo_from_flags(cfg.flags);
        let children = demo.children();
        assert_eq!(children.len(), 2);
        assert_eq!((demo.id(), children[0].id(), children[1].id()), (0, 1, 2));
        let pts = points(cfg.seeded_points);
        if ci > 0 {
            o.push(',');
        }
        let (chs, cm, sr, sm, mdcm, ds) = cfg.params;
        write!(
            o,
            "{{\"flags\":{:?},\"params\":{{\"cube_half_side\":{},\"cube_material\":{},\"sphere_radius\":{},\"sphere_material\":{},\"max_distance_custom_material\":{},\"disable_sphere\":{}}},\"points\":[",
            cfg.flags, hx(chs), cm, hx(sr), sm, hx(mdcm), ds
        )
        .unwrap();
        for (i, p) in pts.iter().enumerate() {
            if i > 0 {
                o.push(',');
            }
            o.push_str(&hx_list(p));
        }
        o.push_str("],\"ids\":{");
        let tree: [(&str, &dyn SDFSurface); 3] = [("0", &demo), ("1", children[0].as_ref()), ("2", children[1].as_ref())];
        for (k, (id, sdf)) in tree.iter().enumerate() {
            if k > 0 {
                o.push(',');
            }
            write!(o, "\"{}\":{{\"bounding_box\":", id).unwrap();
            let bb = sdf.bounding_box();
            o.push_str(&hx_list(&[bb[0].x, bb[0].y, bb[0].z, bb[1].x, bb[1].y, bb[1].z]));
            for (key, distance_only) in [("sample", false), ("sample_distance_only", true)] {
                write!(o, ",\"{}\":[", key).unwrap();
                for (i, p) in pts.iter().enumerate() {
                    if i > 0 {
                        o.push(',');
                    }
                    o.push_str(&sample_words(*sdf, *p, distance_only));
                }
                o.push(']');
            }
            // normal(p, None): the overrides of the demo tree (demo/mod.rs:147-156, cube.rs:164-177, sphere.rs:122-124)
            o.push_str(",\"normal\":[");
            for (i, p) in pts.iter().enumerate() {
                if i > 0 {
                    o.push(',');
                }
                o.push_str(&normal_words(*sdf, *p, None));
            }
            o.push_str("]}");
        }
        o.push_str("}}");
    }
    o.push_str("]}");
    let path = out_dir().join("ref_samples.json");
    std::fs::write(&path, o).expect("write ref_samples.json");
    println!("wrote {}", path.display());
}

/// ref_srgb.json: Srgba::from(Vector3<f32>) then to_linear_srgb() -- the call at scene/sdf/mod.rs:201 -- on values that
/// tell a truncating conversion from a rounding one in one look: c = k/255 moved by {0, +-1 ulp, +0.4/255, +0.6/255},
/// plus the out-of-range and NaN cases.  Records the u8 the conversion produced and the linear value's bits.
#[test]
fn golden_gen_srgb() {
    let mut cases: Vec<f32> = Vec::new();
    for k in 0..=255u32 {
        let c = k as f32 / 255.0;
        cases.push(c);
        cases.push(f32::from_bits(c.to_bits().wrapping_add(1)));
        if k > 0 {
            cases.push(f32::from_bits(c.to_bits() - 1));
        }
        cases.push((k as f32 + 0.4) / 255.0);
        cases.push((k as f32 + 0.6) / 255.0);
    }
    cases.extend_from_slice(&[-0.0, -0.25, 1.5, 256.0, f32::INFINITY, f32::NEG_INFINITY, f32::NAN, 0.5, 0.6, 0.7]);
    let mut o = String::from("{\"generator\":\"sdf-viewer golden_gen v1\",\"cases\":[");
    for (i, c) in cases.iter().enumerate() {
        if i > 0 {
            o.push(',');
        }
        let srgba = Srgba::from(Vector3::new(*c, *c, *c));
        let lin = srgba.to_linear_srgb();
        write!(o, "{{\"c\":{},\"u8\":{},\"linear\":{}}}", hx(*c), srgba.r, hx(lin.x)).unwrap();
    }
    o.push_str("]}");
    let path = out_dir().join("ref_srgb.json");
    std::fs::write(&path, o).expect("write ref_srgb.json");
    println!("wrote {}", path.display());
}

/// An `SDFViewer` for `update()` WITHOUT a GL context.  `update` (scene/sdf/mod.rs:128-217) reads and writes only
/// `tex0`, `tex1`, `loading_mgr`, `bounding_box`, `changed_box` and `changed_box_while_loading`; `volume` and `ctx` (the
/// GPU side, which `new_voxels` needs a three_d::Context for) are never touched by it.  So this test-only helper
/// initialises exactly the fields `update` uses and leaves the two GL fields uninitialised; the value lives in a
/// `MaybeUninit` that is never dropped and on which nothing but `update` is ever called.  (With a display at hand,
/// `SDFViewer::new_voxels(&three_d::HeadlessContext::new().unwrap(), ..)` is the clean way; the textures it makes are the
/// same `texture_from_data(voxels, vec![[AIR_DIST; 4]; n])`.)
fn viewer_without_gl(voxels: Vector3<usize>, bb: [Vector3<f32>; 2], passes: usize) -> Box<MaybeUninit<SDFViewer>> {
    let mut v: Box<MaybeUninit<SDFViewer>> = Box::new(MaybeUninit::uninit());
    let p = v.as_mut_ptr();
    let n = voxels.x * voxels.y * voxels.z;
    unsafe {
        addr_of_mut!((*p).tex0).write(SDFViewer::texture_from_data(voxels, vec![[AIR_DIST; 4]; n]));
        addr_of_mut!((*p).tex1).write(SDFViewer::texture_from_data(voxels, vec![[AIR_DIST; 4]; n]));
        addr_of_mut!((*p).loading_mgr).write(LoadingManager::new(voxels, passes));
        addr_of_mut!((*p).bounding_box).write(bb);
        addr_of_mut!((*p).changed_box).write(None);
        addr_of_mut!((*p).changed_box_while_loading).write(false);
    }
    v
}

fn texture_words(data: &TextureData) -> String {
    match data {
        TextureData::RgbaF32(d) => {
            let mut s = String::from("[");
            for (i, t) in d.iter().enumerate() {
                if i > 0 {
                    s.push(',');
                }
                s.push_str(&hx_list(t));
            }
            s.push(']');
            s
        }
        _ => panic!("expected RgbaF32 texture data"),
    }
}

/// Runs `update` until it reports no work and returns the iterations it consumed.
fn update_to_completion(v: &mut SDFViewer, sdf: &SDFDemo) -> usize {
    let mut total = 0;
    loop {
        let n = v.update(sdf, instant::Duration::from_secs(3600));
        total += n;
        if n == 0 {
            return total;
        }
    }
}

/// ref_grid_9x7x5.json: both textures after SDFViewer::update ran to completion (a LoadingManager of 2 passes) over a
/// 9 x 7 x 5 grid of the demo's bounding box, for the default configuration and for one with the materials swapped;
/// then, on the default one, after a parameter edit
/// (max_distance_custom_material 0.05 -> 0.1 through set_parameter, which makes changed() report the bounding box and
/// update() start its 3-pass manager, scene/sdf/mod.rs:131-156).
#[test]
fn golden_gen_grid() {
    let voxels = Vector3::new(9usize, 7, 5);
    let mut o = String::from("{\"generator\":\"sdf-viewer golden_gen v1\",\"dims\":[9,7,5],\"loading_passes\":2,");
    write!(o, "\"air_dist\":{},\"grids\":[", hx(AIR_DIST)).unwrap();
    for (gi, ci) in [0usize, 1].iter().enumerate() {
        let cfg = &CONFIGS[*ci];
        let demo = demo_from_flags(cfg.flags);
        let bb = demo.bounding_box();
        let mut boxed = viewer_without_gl(voxels, bb, 2);
        let v: &mut SDFViewer = unsafe { &mut *boxed.as_mut_ptr() };
        let iterations = update_to_completion(v, &demo);
        if gi > 0 {
            o.push(',');
        }
        write!(o, "{{\"config\":{},\"iterations\":{},\"passes_left\":{},\"tex0\":", ci, iterations, v.loading_mgr.passes_left()).unwrap();
        o.push_str(&texture_words(&v.tex0.data));
        o.push_str(",\"tex1\":");
        o.push_str(&texture_words(&v.tex1.data));
        if *ci == 0 {
            // the edit: demo/mod.rs:119-133 (parameter id 0 = max_distance_custom_material)
            demo.set_parameter(0, &SDFParamValue::Float(0.1)).expect("set_parameter");
            let edit_iterations = update_to_completion(v, &demo);
            write!(o, ",\"edit\":{{\"max_distance_custom_material\":{},\"iterations\":{},\"tex0\":", hx(0.1), edit_iterations).unwrap();
            o.push_str(&texture_words(&v.tex0.data));
            o.push_str(",\"tex1\":");
            o.push_str(&texture_words(&v.tex1.data));
            o.push('}');
        }
        o.push('}');
        std::mem::forget(boxed); // never dropped: two of its fields were never initialised
    }
    o.push_str("]}");
    let path = out_dir().join("ref_grid_9x7x5.json");
    std::fs::write(&path, o).expect("write ref_grid_9x7x5.json");
    println!("wrote {}", path.display());
}
