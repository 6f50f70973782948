"""Row tiling of one frame over the GPUs of a node (host logic, no device code).

Image rows are dealt to the ranks in bands of BAND rows: band b belongs to rank b % n_parts.  Every
rank renders its bands into one compact buffer [part_rows, width, ...]; rank 0 gathers the buffers
(RCCL: each rank sends straight to rank 0 over its own xGMI link, no ring) and de-interleaves.
The scene is replicated, so there is no other exchange.
"""
import math

BAND = 8


def frame_size(n_gpus, base=(1920, 1080)):
    """Weak scaling: the frame grows by sqrt(n) per dimension so the rays per GPU stay constant;
    width is a multiple of 8, height a multiple of BAND * n_gpus."""
    s = math.sqrt(n_gpus)
    w = int(round(base[0] * s / 8)) * 8
    q = BAND * n_gpus
    h = int(round(base[1] * s / q)) * q
    return w, h


def part_rows(height, n_parts):
    assert height % (BAND * n_parts) == 0
    return height // n_parts


def rows_of_part(height, n_parts, part, band=BAND):
    """Global row index of every compact row of `part`, in compact order."""
    out = []
    for yc in range(height // n_parts):
        out.append(((yc // band) * n_parts + part) * band + yc % band)
    return out


def deinterleave(parts, band=BAND):
    """parts: list (one per rank, in rank order) of arrays/tensors [part_rows, width, ...] ->
    [height, width, ...] in image row order."""
    n = len(parts)
    first = parts[0]
    pr = first.shape[0]
    nb = pr // band
    if hasattr(first, "new_empty"):     # torch tensors
        import torch
        stacked = torch.stack([p.reshape(nb, band, *p.shape[1:]) for p in parts], dim=1)
    else:
        import numpy as np
        stacked = np.stack([p.reshape(nb, band, *p.shape[1:]) for p in parts], axis=1)
    return stacked.reshape(pr * n, *first.shape[1:])


def gather_to_rank0(buf, world, rank, dist, out=None):
    """torch.distributed gather of equally sized per-rank buffers to rank 0 (list on rank 0, None
    elsewhere).  `out`: optional preallocated receive list on rank 0, reused across frames."""
    if world == 1:
        return [buf]
    import torch
    if rank == 0 and out is None:
        out = [torch.empty_like(buf) for _ in range(world)]
    dist.gather(buf, out if rank == 0 else None, dst=0)
    return out if rank == 0 else None


def broadcast_scene(build, rank, dist, torch, device=None):
    """One scene for all ranks, built once: rank 0 calls build() -> a raytracer_amd.Scene with its tree built and
    broadcasts the flat image of it (rt_scene_export_flat); every rank gets the image back as a uint8 array ready
    for Scene.import_flat / Context.upload_flat.  The reference builds its tree once before its threads fan out
    (main.cpp:678, 735); here the other ranks neither parse meshes nor build anything.  `device`: where the
    collective's tensors live (a CUDA device for the RCCL backend, None for gloo)."""
    import numpy as np

    from . import aligned_bytes
    img = None
    failure = None
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == 0:
        try:
            img = build().export_flat()
            n[0] = img.size
        except Exception as e:      # (the other ranks are waiting in the collective below: tell them instead of leaving them there)
            failure = e
            n[0] = -1
    dist.broadcast(n, src=0)
    size = int(n.item())
    if size < 0:
        raise RuntimeError("rank 0 could not build the scene") from failure
    if rank == 0:
        t = torch.from_numpy(img)
    else:
        img = aligned_bytes(size)
        t = torch.from_numpy(img)
    if device is not None:
        d = t.to(device)
        dist.broadcast(d, src=0)
        if rank != 0:
            t.copy_(d)
        del d
    else:
        dist.broadcast(t, src=0)
    return img
