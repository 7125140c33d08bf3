"""rttnw_amd/host/rttnw — the reference's command line (main.rs:236-258) as a native program: scenes.cpp + main.cpp over
the C ABI, with its own zlib-based PNG codec in place of the reference's `image` crate."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rttnw_amd", "host", "rttnw")


@pytest.fixture(scope="module")
def exe():
    subprocess.run(["make", "-C", os.path.join(ROOT, "rttnw_amd", "csrc")], check=True, capture_output=True)   # the program links against it
    subprocess.run(["make", "-C", os.path.join(ROOT, "rttnw_amd", "host"), "cli"], check=True, capture_output=True)
    return EXE


def test_usage_and_unknown_scene_like_the_reference(exe):
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr and "\t- 9: final_scene" in r.stderr          # main.rs:238-250
    r = subprocess.run([exe, "12"], capture_output=True, text=True)
    assert r.returncode == 1 and "There is no scene 12" in r.stderr and "Scene number: 12" in r.stdout   # main.rs:179-182,253
    r = subprocess.run([exe, "seven"], capture_output=True, text=True)
    assert r.returncode == 1 and "There was an error" in r.stderr                                    # the parse error, main.rs:252


def test_png_codec_round_trip(exe, tmp_path):
    """The decoder that stands in for `image::open` reads assets/earth.png as PIL does; the encoder's files read back equal."""
    from PIL import Image
    src = os.path.join(ROOT, "rttnw_amd", "assets", "earth.png")
    out = tmp_path / "again.png"
    assert subprocess.run([exe, "--reencode", src, str(out)]).returncode == 0
    a = np.asarray(Image.open(src).convert("RGBA"))
    b = np.asarray(Image.open(out))
    assert b.shape == a.shape and b.dtype == np.uint8 and np.array_equal(a, b)
    rgb = tmp_path / "rgb.png"                       # an RGB file with every filter type PIL cares to use
    Image.fromarray(a[..., :3].copy(), "RGB").save(rgb, optimize=True)
    assert subprocess.run([exe, "--reencode", str(rgb), str(out)]).returncode == 0
    assert np.array_equal(np.asarray(Image.open(out))[..., :3], a[..., :3]) and (np.asarray(Image.open(out))[..., 3] == 255).all()
    assert subprocess.run([exe, "--reencode", __file__, str(out)]).returncode == 1


def test_no_device_means_loud_failure(exe, tmp_path):
    from rttnw_amd import library
    if library.product().device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "7", "--width", "16", "--spp", "2", "--out", str(tmp_path / "i.png")], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr and not (tmp_path / "i.png").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("number,name", [(7, "cornell_box"), (4, "earth")])
def test_native_cli_writes_the_oracles_image(exe, gpu, oracle, scenes_lib, earth, tmp_path, number, name):
    import util
    from oracle import rto
    from PIL import Image
    out = tmp_path / "image.png"
    r = subprocess.run([exe, str(number), "--width", "48", "--spp", "6", "--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Scene number: %d" % number in r.stdout and "Running scene %s" % name in r.stdout
    got = np.asarray(Image.open(out))
    so, setup = util.build(oracle, scenes_lib, name, earth)
    h = int(48 / (setup.width / setup.height))
    assert got.shape == (h, 48, 4) and (got[..., 3] == 255).all()
    cam, p = util.params_for(setup, 48, h, 6)
    _, ro, _ = rto.render(so, cam, p)
    assert (got == ro).all(axis=2).mean() >= 0.999
    if number == 7:   # all GPUs of a node in one call (here: three logical ranks on one device): the same image
        out3 = tmp_path / "image3.png"
        r = subprocess.run([exe, "7", "--width", "48", "--spp", "6", "--gpus", "3", "--out", str(out3)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert np.array_equal(np.asarray(Image.open(out3)), got)
