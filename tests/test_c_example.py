"""examples/cornell_box.c — the C ABI from plain C (what a cgo / JNI / Rust-FFI host does): it compiles as C99 against
include/rttnw_hip.h alone, fails loudly without a device, and on an MI355X writes the image the oracle computes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "rttnw_amd", "csrc")


def build(tmp_path):
    exe = str(tmp_path / "cornell_box_c")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "cornell_box.c"), "-L" + LIBDIR, "-lrttnw_hip", "-Wl,-rpath," + LIBDIR, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def read_ppm(path):
    data = open(path, "rb").read()
    magic, dims, maxv, rest = data.split(b"\n", 3)
    w, h = (int(x) for x in dims.split())
    assert magic == b"P6" and maxv == b"255" and len(rest) == w * h * 3
    return np.frombuffer(rest, dtype=np.uint8).reshape(h, w, 3)


def test_c_caller_builds_and_fails_loudly_without_a_device(tmp_path):
    from rttnw_amd import library
    exe = build(tmp_path)
    if library.product().device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "16", "2", str(tmp_path / "x.ppm")], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr and not (tmp_path / "x.ppm").exists()
    assert subprocess.run([exe, "0"], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_c_caller_writes_the_oracles_image(gpu, oracle, scenes_lib, tmp_path):
    import util
    from oracle import rto
    exe = build(tmp_path)
    out = tmp_path / "c.ppm"
    r = subprocess.run([exe, "64", "8", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = read_ppm(out)
    so, setup = util.build(oracle, scenes_lib, "cornell_box")
    cam, p = util.params_for(setup, 64, 64, 8)
    _, ro, _ = rto.render(so, cam, p)
    assert (got == ro[..., :3]).all(axis=2).mean() >= 0.999
    # the node-level entry point with three logical ranks on this GPU: the same file
    out3 = tmp_path / "c3.ppm"
    r = subprocess.run([exe, "64", "8", str(out3), "f64", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out3, "rb").read() == open(out, "rb").read()
