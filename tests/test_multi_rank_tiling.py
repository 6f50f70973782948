"""N > 1 path on CPU: two `gloo` ranks each produce the hit ids of their row bands, rank 0 gathers
and de-interleaves, and the result must equal the single-process frame.  The per-rank pixels come
from the CPU oracle here (no GPU in this test); on the GPU box the same tiling functions carry the
device buffers (bench.py)."""
import os
import sys

import numpy as np
import pytest

from conftest import CAM, ROOT, build_scene
from raytracer_amd import tiling


def test_tiling_arithmetic():
    for n in (1, 2, 4, 8):
        w, h = tiling.frame_size(n)
        assert w % 8 == 0 and h % (tiling.BAND * n) == 0
        rays_per_gpu = w * h / n
        assert abs(rays_per_gpu / (1920 * 1080) - 1) < 0.02
        seen = sorted(r for part in range(n) for r in tiling.rows_of_part(h, n, part))
        assert seen == list(range(h))
    assert tiling.frame_size(1) == (1920, 1080)
    parts = [np.arange(4 * 8 * 3).reshape(32, 3) * 0 + p for p in range(2)]
    full = tiling.deinterleave(parts)
    assert full.shape == (64, 3)
    assert [int(full[r, 0]) for r in (0, 7, 8, 15, 16, 63)] == [0, 0, 1, 1, 0, 1]


def _worker(rank, world, port, w, h, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle.bindings import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = build_scene(Oracle(), "teapot_normal.obj")
    org, d = orc.primary_rays(w, h, CAM)
    rows = tiling.rows_of_part(h, world, rank)
    idx = (np.array(rows)[:, None] * w + np.arange(w)[None, :]).ravel()
    ids, _ = orc.shoot(org[idx], d[idx])
    buf = torch.from_numpy(ids.reshape(len(rows), w).copy())
    parts = tiling.gather_to_rank0(buf, world, rank, dist)
    if rank == 0:
        np.save(out_path, tiling.deinterleave(parts).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_composes_the_frame(tmp_path):
    import torch.multiprocessing as mp
    from oracle.bindings import Oracle
    world, w, h = 2, 96, 64
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, w, h, out), nprocs=world, join=True)
    orc = build_scene(Oracle(), "teapot_normal.obj")
    org, d = orc.primary_rays(w, h, CAM)
    ids, _ = orc.shoot(org, d)
    assert np.array_equal(np.load(out), ids.reshape(h, w))


def _scene_worker(rank, world, port, out_dir):
    """Rank 0 loads the mesh and builds the tree; the others receive the flat image and never see the mesh file."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import raytracer_amd as rt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    build = (lambda: build_scene(rt.Scene(), "water.obj")) if rank == 0 else None
    img = tiling.broadcast_scene(build, rank, dist, torch)
    s = rt.Scene()
    s.import_flat(img)
    rec, boxes = s.bvh_dump()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rec=rec, boxes=boxes, n=s.size(), img=np.asarray(img))
    dist.barrier()
    dist.destroy_process_group()


def test_scene_built_once_and_broadcast(tmp_path):
    """One build, N uploads: the flat image broadcast from rank 0 gives every rank the reference's tree."""
    import torch.multiprocessing as mp
    from conftest import golden
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_scene_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = golden("bvh_water.npz")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for r in (r0, r1):
        assert np.array_equal(r["rec"], g["rec"]) and np.array_equal(r["boxes"], g["boxes"]) and int(r["n"]) == 5008
    assert np.array_equal(r0["img"], r1["img"])
