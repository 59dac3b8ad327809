//! Headless renderer: `headless <scene.yaml | scene.gltf> [--spp N] [--out frame.png] [--bounces B] [--device D]`.
//! The reference's `main` takes only a positional scene path and opens a window (src/main.rs:163-181); this binary
//! drives the same `Scene` through `HipRenderer` without one. Not compiled in the build image (no Rust toolchain);
//! `tools/render.py` is the same program over the Python facade and runs in the tests.
use std::env;
use std::path::PathBuf;
use std::time::Instant;

use anyhow::{bail, Result};

use ray_tracer::hip::HipRenderer;
use ray_tracer::scene::Scene;

fn main() -> Result<()> {
    pretty_env_logger::init();
    let mut scene_path: Option<String> = None;
    let (mut spp, mut bounces, mut device) = (64u32, 24u32, 0i32); // BOUNCES is 24 upstream (src/shaders/main.comp:5)
    let mut out = PathBuf::from("frame.png");
    let mut args = env::args().skip(1);
    while let Some(a) = args.next() {
        let mut value = |name: &str| args.next().ok_or_else(|| anyhow::anyhow!("{name} needs a value"));
        match a.as_str() {
            "--spp" => spp = value("--spp")?.parse()?,
            "--bounces" => bounces = value("--bounces")?.parse()?,
            "--device" => device = value("--device")?.parse()?,
            "--out" => out = PathBuf::from(value("--out")?),
            _ if a.starts_with("--") => bail!("unknown option {a}"),
            _ => scene_path = Some(a),
        }
    }
    let scene_path = scene_path.unwrap_or_else(|| "./scenes/nice/lego_bulldozer.yaml".into()); // main.rs:166-168
    let scene = if scene_path.ends_with(".gltf") || scene_path.ends_with(".glb") {
        Scene::from_gltf(&scene_path)? // scene/gltf.rs:131
    } else {
        Scene::from_new(&scene_path)? // scene/loader.rs:11
    };

    let mut renderer = HipRenderer::create(scene, device, bounces)?;
    let report = renderer.tune_resident_set()?;
    log::info!(
        "node visits served from LDS: {:.0} % -> {:.0} %",
        100.0 * report.resident_share_before[1],
        100.0 * report.resident_share_after[1]
    );
    let start = Instant::now();
    renderer.dispatch_samples(spp)?; // one call: the library batches the samples into large wavefront passes
    renderer.wait_idle()?;
    let secs = start.elapsed().as_secs_f64();
    let (w, h) = renderer.render_resolution();
    log::info!("{spp} spp at {w}x{h} in {secs:.3} s ({:.1} samples/s)", spp as f64 / secs);
    renderer.save_frame(&out)?;
    Ok(())
}
