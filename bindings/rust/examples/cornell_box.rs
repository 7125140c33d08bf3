//! scenes.rs:157-196 + main.rs:137-150 of luliic2/rttnw, written against the wrapper.  `cargo run --example cornell_box`
//! writes image.png like `cargo run --release -- 7` of the reference does.  (Un-compiled here: no Rust toolchain.)
use rttnw_hip::scene::{reference_camera, reference_params, Plane, SceneBuilder};

fn main() -> Result<(), rttnw_hip::scene::Error> {
    let mut b = SceneBuilder::new(0x5eed_0001)?;
    let red = { let t = b.solid([0.65, 0.05, 0.05])?; b.lambertian(t)? };
    let white = { let t = b.solid([0.73, 0.73, 0.73])?; b.lambertian(t)? };
    let green = { let t = b.solid([0.12, 0.45, 0.15])?; b.lambertian(t)? };
    let light = { let t = b.solid([15., 15., 15.])?; b.diffuse_light(t)? };
    let world = b.list()?;
    for h in [
        b.rectangle(Plane::YZ, green, 0. ..555., 0. ..555., 555.)?,
        b.rectangle(Plane::YZ, red, 0. ..555., 0. ..555., 0.)?,
        b.rectangle(Plane::XZ, light, 213. ..343., 227. ..332., 554.)?,
        b.rectangle(Plane::XZ, white, 0. ..555., 0. ..555., 0.)?,
        b.rectangle(Plane::XZ, white, 0. ..555., 0. ..555., 555.)?,
        b.rectangle(Plane::XY, white, 0. ..555., 0. ..555., 555.)?,
    ] {
        b.push(world, h)?;
    }
    // scenes.rs:180-188: Cube::new(..).rotate_y(15.).translate(..)
    let tall = b.cube([0., 0., 0.], [165., 330., 165.], white)?;
    let tall = b.rotate_y(tall, 15.)?;
    let tall = b.translate(tall, [265., 0., 295.])?;
    b.push(world, tall)?;
    let small = b.cube([0., 0., 0.], [165., 165., 165.], white)?;
    let small = b.rotate_y(small, -18.)?;
    let small = b.translate(small, [130., 0., 65.])?;
    b.push(world, small)?;
    let scene = b.commit(world)?;

    let (width, height, samples) = (600u32, 600u32, 200u32); // main.rs:137-141
    let cam = reference_camera([278., 278., -800.], [278., 278., 0.], 40., 1.0, 0.0);
    let p = reference_params(width, height, samples, [0., 0., 0.]);
    let (rgba, stats) = scene.render(&cam, &p)?;
    println!("{} samples in {:.1} ms of device time", stats.samples, stats.kernel_ms);
    image::save_buffer("image.png", &rgba, width, height, image::ColorType::Rgba8).unwrap();
    Ok(())
}
