"""N > 1 host path on CPU: world_size-2 gloo.  Each rank produces ITS tiles (with the oracle as the
tile renderer — there is no GPU here), packs them in the ABI's packed order, rank 0 gathers with
torch.distributed and un-tiles; the result must equal the single-rank image bit for bit (keyed RNG +
per-pixel ownership make the image independent of the partition)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import torch
    import torch.distributed as dist
    import util
    from oracle import rto
    from rttnw_amd import abi, tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = rto.binding()
    scenes = abi.Binding(C.CDLL(os.path.join(ROOT, "rttnw_amd", "host", "librttnw_scenes.so")), "", abi.SCENES_FUNCS)
    sc, setup = util.build(b, scenes, "cornell_box")
    w, h = 45, 37
    cam, p = util.params_for(setup, w, h, 4, spp_chunk=2, tile_rank=rank, tile_world=world)
    lin, _, _ = rto.render(sc, cam, p, n_threads=2)               # only this rank's tiles are written
    lay = tiles.layout(w, h, world)
    packed = torch.from_numpy(tiles.pack_rank(lin, rank, world))
    assert packed.shape == (lay["pixels_per_rank"], 4)
    glist = [torch.zeros_like(packed) for _ in range(world)] if rank == 0 else None
    dist.gather(packed, glist, dst=0)                              # same call pattern as DeviceRenderer.collect
    if rank == 0:
        img = tiles.untile_reference(torch.stack(glist).numpy(), w, h, world)
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gather_equals_single_rank(tmp_path, oracle, scenes_lib):
    import torch.multiprocessing as mp
    import util
    from oracle import rto
    out = str(tmp_path / "img.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    sc, setup = util.build(oracle, scenes_lib, "cornell_box")
    cam, p = util.params_for(setup, 45, 37, 4, spp_chunk=2)
    want, _, _ = rto.render(sc, cam, p)
    assert np.array_equal(got, want)


def _gpu_worker(rank, world, port, out_path):
    """One process per rank, as bench.py runs them — here both on cuda:0, with gloo carrying the gather (a 1-GPU box
    cannot run two RCCL ranks): the device kernels trace the rank's tiles, rank 0 un-tiles on the device."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import torch
    import torch.distributed as dist
    import util
    from rttnw_amd import abi, library, render
    from rttnw_amd.abi import check
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    gpu = library.product()
    sc, setup = util.build(gpu, library.scenes(), "final_scene", None)
    w, h = 100, 76  # ragged: 13 x 10 tiles, the last column / row partly outside the image
    cam, p = util.params_for(setup, w, h, 6, precision=abi.F32, tile_rank=rank, tile_world=world, seed=4)
    r = render.DeviceRenderer(sc, cam, p)
    r.trace()
    torch.cuda.synchronize()
    packed = r.packed.cpu()
    glist = [torch.zeros_like(packed) for _ in range(world)] if rank == 0 else None
    dist.gather(packed, glist, dst=0)
    if rank == 0:
        gathered = torch.stack(glist).cuda()
        stream = torch.cuda.current_stream().cuda_stream
        check(gpu.untile_device(w, h, world, p.precision, gathered.data_ptr(), r.linear.data_ptr(), r.rgba8.data_ptr(), stream),
              gpu, "rttnw_untile_device")
        torch.cuda.synchronize()
        np.save(out_path, r.linear.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_multi_process_device_ranks_equal_single_rank(tmp_path, gpu, scenes_lib, world):
    import torch.multiprocessing as mp
    import util
    from rttnw_amd import abi, render
    out = str(tmp_path / "img.npy")
    mp.spawn(_gpu_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    sc, setup = util.build(gpu, scenes_lib, "final_scene", None)
    cam, p = util.params_for(setup, 100, 76, 6, precision=abi.F32, seed=4)
    want, _, _ = render.render_host(sc, cam, p)
    assert np.array_equal(got.astype(np.float64), want)
