"""`python -m rttnw_amd <scene>` — the reference's CLI (src/main.rs:236-258) on the HIP path.

One positional argument, the scene number 1..9 with the reference's per-scene defaults (size, spp, camera:
main.rs:66-183); writes `image.png` into the current directory (main.rs:231) and prints the wall time.
Optional extras (not in the reference): --spp, --width, --precision f32|f64, --out, --seed, --passes.
"""
import argparse
import sys
import time

import numpy as np

USAGE = """Usage: python -m rttnw_amd <scene>
Possible scenes:
\t- 1: random_scene
\t- 2: two_spheres
\t- 3: two_perlin_spheres
\t- 4: earth
\t- 5: simple_light
\t- 6: empty_cornell_box
\t- 7: cornell_box
\t- 8: smoke_cornell_box
\t- 9: final_scene"""


def main(argv=None):
    ap = argparse.ArgumentParser(add_help=True, usage=USAGE)
    ap.add_argument("scene", type=int)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--precision", default="f64", choices=["f32", "f64", "f64strict"], help="f64 = the reference's arithmetic (default); f32 = throughput; f64strict = f64 with nothing contracted (bit-for-bit the CPU reference's path decisions)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="image.png")
    ap.add_argument("--passes", type=int, default=1, help="render in this many passes over disjoint sample ranges, "
                    "rewriting the image after each (progressive)")
    try:
        args = ap.parse_args(argv)
    except SystemExit:
        print("There was an error", file=sys.stderr)   # DummyError, main.rs:260-268
        raise
    from PIL import Image
    from . import abi, library, render
    from .abi import CameraDesc
    from .scene import Scene, load_earth, make_params

    scenes = library.scenes()
    name = scenes.scenes_name(args.scene)
    if not name:
        print("There is no scene %d" % args.scene, file=sys.stderr)   # main.rs:179-182
        return 1
    print("Scene number: %d" % args.scene)
    print("Running scene %s" % name.decode())
    t0 = time.time()
    sc = Scene(library.product(), scenes_binding=scenes)
    setup = sc.build_named(name.decode(), earth_rgba=load_earth())
    w = args.width or setup.width
    h = int(w / (setup.width / setup.height))                        # height = (width / aspect) as u32, main.rs:184
    cam = CameraDesc.from_buffer_copy(setup.camera)
    cam.aspect_ratio = setup.width / setup.height
    p = make_params(w, h, args.spp or setup.spp, background=tuple(setup.background), seed=args.seed,
                    precision={"f32": abi.F32, "f64": abi.F64, "f64strict": abi.F64_STRICT}[args.precision])
    if args.passes > 1:
        def show(k, linear):
            Image.fromarray(render.quantise_rgba8(linear), "RGBA").save(args.out)
            print("pass %d/%d written" % (k + 1, args.passes))
        _, rgba, _ = render.render_host_passes(sc, cam, p, args.passes, on_pass=show)
        print("%.3fs" % (time.time() - t0))
        return 0
    _, rgba, st = render.render_host(sc, cam, p)
    Image.fromarray(np.ascontiguousarray(rgba), "RGBA").save(args.out)
    print("%.3fs (trace kernel %.1f ms, %.1f Msamples/s)" % (time.time() - t0, st.kernel_ms,
                                                          st.samples / max(st.kernel_ms, 1e-9) / 1e3))
    return 0


if __name__ == "__main__":
    sys.exit(main())
