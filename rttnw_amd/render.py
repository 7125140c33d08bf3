"""Render drivers over the C ABI: the `render()` of the reference's main.rs:58-233, minus PNG/progress.

PyTorch is plumbing here (device buffers, the current HIP stream, torch.distributed over RCCL);
all tracing happens in the hand-written HIP kernels behind `rttnw_render_tiles_device`.
"""
import ctypes as C

import numpy as np

from . import abi, library, tiles
from .abi import Stats, check


def render_host(scene, cam, params, want_stats=True):
    """Blocking single-GPU render with host outputs: (linear HxWx3 f64, rgba8 HxWx4 u8, Stats)."""
    b = library.product()
    h, w = params.height, params.width
    lin = np.zeros((h, w, 3), dtype=np.float64)
    rgba = np.zeros((h, w, 4), dtype=np.uint8)
    st = Stats()
    rc = b.render(scene.handle, C.byref(cam), C.byref(params), lin.ctypes.data, rgba.ctypes.data,
                  C.byref(st) if want_stats else None)
    check(rc, b, "rttnw_render")
    return lin, rgba, st


def render_multi(scene, cam, params, device_ids, want_stats=True):
    """`rttnw_render_multi`: one call, the GPUs (or logical ranks) of `device_ids`; (linear, rgba8, [Stats per rank])."""
    b = library.product()
    h, w = params.height, params.width
    lin = np.zeros((h, w, 3), dtype=np.float64)
    rgba = np.zeros((h, w, 4), dtype=np.uint8)
    n = len(device_ids)
    ids = (C.c_int32 * n)(*device_ids)
    st = (Stats * n)()
    rc = b.render_multi(scene.handle, C.byref(cam), C.byref(params), n, ids, lin.ctypes.data, rgba.ctypes.data,
                        C.cast(st, C.c_void_p) if want_stats else None)
    check(rc, b, "rttnw_render_multi")
    return lin, rgba, list(st)


def render_host_passes(scene, cam, params, passes, on_pass=None):
    """The same image as `render_host`, in `passes` passes over disjoint sample ranges (`rttnw_params.sample_begin`):
    after every pass the running mean is a complete, displayable estimate — progressive display and a natural
    checkpoint (persist the running sum and the next sample index) for long renders.  `on_pass(k, linear)` is called
    with the running mean after pass k.  Returns (linear HxWx3 f64, rgba8 HxWx4 u8, samples per pixel done)."""
    import copy
    total, done = None, 0
    base, spp = params.sample_begin, params.spp
    for k in range(passes):
        n = spp // passes + (1 if k < spp % passes else 0)
        if n == 0:
            continue
        p = copy.copy(params)
        p.spp, p.sample_begin = n, base + done
        lin, _, _ = render_host(scene, cam, p, want_stats=False)
        total = lin * n if total is None else total + lin * n
        done += n
        if on_pass is not None:
            on_pass(k, total / done)
    mean = total / max(done, 1)
    return mean, quantise_rgba8(mean), done


def quantise_rgba8(linear):
    """main.rs:219-225 on a linear image: sqrt, clamp to 0.999, * 256, `as u8`; alpha 255."""
    x = np.sqrt(np.maximum(linear, 0.0))
    x = np.minimum(x, 0.999) * 256.0
    rgba = np.full(linear.shape[:2] + (4,), 255, dtype=np.uint8)
    rgba[..., :3] = np.nan_to_num(x, nan=0.0).astype(np.uint8)
    return rgba


def _torch_dtype(precision):
    import torch
    return torch.float32 if precision == abi.F32 else torch.float64


class DeviceRenderer:
    """Device-resident render of this rank's tiles (+ gather over torch.distributed for world > 1).

    Buffers are torch tensors on the current device; kernels are launched on torch's current stream.
    """

    def __init__(self, scene, cam, params, group=None):
        import torch
        self.torch = torch
        self.b = library.product()
        self.scene, self.cam, self.params = scene, cam, params
        self.world = params.tile_world
        self.rank = params.tile_rank
        self.group = group
        self.lay = tiles.layout(params.width, params.height, self.world)
        dt = _torch_dtype(params.precision)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.packed = torch.zeros((self.lay["pixels_per_rank"], 4), dtype=dt, device=dev)
        root = self.rank == 0
        # force_gather: run the collective with a single rank too (launch-plumbing check on a 1-GPU box)
        self.force_gather = bool(group is None and self.world == 1 and __import__("os").environ.get("RTTNW_BENCH_FORCE_DIST") == "1")
        self.gathered = (torch.zeros((self.world, self.lay["pixels_per_rank"], 4), dtype=dt, device=dev)
                         if (root and (self.world > 1 or self.force_gather)) else None)
        self.linear = torch.zeros((params.height, params.width, 3), dtype=dt, device=dev) if root else None
        self.rgba8 = torch.zeros((params.height, params.width, 4), dtype=torch.uint8, device=dev) if root else None

    def trace(self, stats=None):
        """Launch the trace + resolve kernels for this rank's tiles (asynchronous)."""
        stream = self.torch.cuda.current_stream().cuda_stream
        rc = self.b.render_tiles_device(self.scene.handle, C.byref(self.cam), C.byref(self.params),
                                        self.packed.data_ptr(), stream, C.byref(stats) if stats is not None else None)
        check(rc, self.b, "rttnw_render_tiles_device")

    def collect(self):
        """Gather every rank's packed tiles on rank 0 (RCCL over xGMI) and scatter them into the framebuffer."""
        src = self.packed
        if self.world > 1 or self.force_gather:
            import torch.distributed as dist
            if dist.get_backend(self.group) == "gloo":
                # rehearsal of an N-rank run on ONE device (bench.py RTTNW_BENCH_ONE_DEVICE=1; RCCL refuses two ranks on a device):
                # gloo gathers host tensors only, so the packed tiles go through the host
                host = self.packed.cpu()
                hlist = [self.torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(host, hlist, dst=0, group=self.group)
                if self.rank == 0:
                    self.gathered.copy_(self.torch.stack(hlist))
            else:
                glist = list(self.gathered.unbind(0)) if self.rank == 0 else None
                dist.gather(self.packed, glist, dst=0, group=self.group)   # RCCL: grouped send/recv into rank 0
            src = self.gathered
        if self.rank == 0:
            p = self.params
            stream = self.torch.cuda.current_stream().cuda_stream
            rc = self.b.untile_device(p.width, p.height, self.world, p.precision, src.data_ptr(),
                                      self.linear.data_ptr(), self.rgba8.data_ptr(), stream)
            check(rc, self.b, "rttnw_untile_device")

    def step(self):
        self.trace()
        self.collect()
