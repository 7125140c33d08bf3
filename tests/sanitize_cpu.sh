#!/bin/bash
# CPU sanitizer runs of the product's host-side code (the lowering with its host threads, the C-ABI builder, the tracing core as
# tests/hostsim builds it).  GPU AddressSanitizer is not available on the MI355X pool, so this is where the sanitizers run.
#   bash tests/sanitize_cpu.sh            # ThreadSanitizer on a 200 000-sphere commit, then ASan + UBSan on five scenes x 2 precisions
# Not part of the pytest suite (two instrumented builds + runs: ~4 min on 8 cores).  Exit status 0 = no report.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC="$ROOT/tests/hostsim/hostsim.cpp $ROOT/rttnw_amd/csrc/capi_builder.cpp $ROOT/rttnw_amd/csrc/scene_lower.cpp"
OUT=${TMPDIR:-/tmp}/rttnw_sanitize; mkdir -p $OUT
make -s -C $ROOT/rttnw_amd/host librttnw_scenes.so
cat > $OUT/run.py <<PY
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
from rttnw_amd import abi, scene as S
import util
lib = C.CDLL(sys.argv[1])
b = abi.Binding(lib, "rttnw_", abi.BUILDER_FUNCS)
b.add([("builder", C.c_void_p, []),
       ("debug_scene_nodes", C.c_int, [abi.scene_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]),
       ("debug_scene_nodes4", C.c_int, [abi.scene_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)])])
lib.hostsim_render.restype = C.c_int
lib.hostsim_render.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc), C.POINTER(abi.Params), C.c_void_p, C.POINTER(abi.Stats), C.c_int]
sl = abi.Binding(C.CDLL("$ROOT/rttnw_amd/host/librttnw_scenes.so"), "", abi.SCENES_FUNCS)
earth = S.load_earth()
cases = [("spheres_1m", 200000)] if sys.argv[2] == "commit" else [("final_scene", 0), ("cornell_box", 0), ("smoke_cornell_box", 0), ("random_scene", 0), ("spheres_1m", 150000)]
for name, n in cases:
    sc, setup = util.build(b, sl, name, earth, n)
    if sys.argv[2] == "commit": continue
    for prec in (abi.F64, abi.F32):
        cam, p = util.params_for(setup, 24, 24, 3, precision=prec, seed=3, collect_counters=1)
        os.environ.pop("HOSTSIM_QUANT", None)
        img, st = util.hostsim_render(b, sc, cam, p)
        for mode in ("1",) + (("3",) if n == 0 else ()):   # ... and through the quantised records of the decoupled kernels (bvh_quant.hpp); small scenes: the walk that never culls
            os.environ["HOSTSIM_QUANT"] = mode
            img_q, st_q = util.hostsim_render(b, sc, cam, p)
            assert np.array_equal(img, img_q) or (mode == "3" and prec == abi.F32)
    if n == 0 and name in ("final_scene", "cornell_box"):      # rays nothing can cull (NaN direction: lookfrom == lookat) must stay inside the stack (round-4 advisor)
        for prec in (abi.F64, abi.F32):
            cam, p = util.params_for(setup, 16, 16, 2, precision=prec, seed=3)
            for k in range(3):
                cam.lookat[k] = cam.lookfrom[k]
            os.environ.pop("HOSTSIM_QUANT", None)
            util.hostsim_render(b, sc, cam, p)
            os.environ["HOSTSIM_QUANT"] = "1"
            util.hostsim_render(b, sc, cam, p)
    if n == 0 and name in ("final_scene", "cornell_box"):      # the lockstep wave model (experiment support) on a few jobs
        out = np.zeros(64, dtype=np.uint64)
        cam, p = util.params_for(setup, 64, 64, 8, precision=abi.F64, seed=3)
        lib.hostsim_wave_model.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc), C.POINTER(abi.Params), C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
        for policy, a, bb in ((0, 2, 0), (1, 0, 128), (2, 16, 8), (4, 2, 56)):
            lib.hostsim_wave_model(sc.handle, C.byref(cam), C.byref(p), policy, a, bb, 2, 128, out.ctypes.data)
    print(name, "ok", flush=True)
PY
FLAGS="-O1 -g -std=c++17 -fPIC -pthread -Wno-unknown-pragmas -shared"
g++ $FLAGS -fsanitize=thread -o $OUT/libhostsim_tsan.so $SRC
LD_PRELOAD=$(g++ -print-file-name=libtsan.so) TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66" python $OUT/run.py $OUT/libhostsim_tsan.so commit 2> $OUT/tsan.log || { grep -m5 -A12 "WARNING: ThreadSanitizer" $OUT/tsan.log; echo "ThreadSanitizer: reports in $OUT/tsan.log"; exit 1; }
echo "ThreadSanitizer: clean (parallel collect / SAH splits / record emission of 200 000 spheres)"
g++ $FLAGS -fsanitize=address,undefined -fno-sanitize-recover=undefined -o $OUT/libhostsim_asan.so $SRC
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python $OUT/run.py $OUT/libhostsim_asan.so render 2> $OUT/asan.log || { tail -30 $OUT/asan.log; exit 1; }
echo "AddressSanitizer + UBSan: clean"
