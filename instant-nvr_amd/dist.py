"""Multi-GPU frame rendering: rays shard by image tile across the ranks of one node
(one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU
tests).  Every rank holds a full replica of the hash tables / MLPs and of the per-frame scene
tensors; the only exchange is ONE all-gather of the rendered [r,g,b,acc] tiles
(262,144 rays x 16 B = 4 MB per 512x512 frame).  Tiles are dealt cyclically (round-robin) because
contiguous bands are load-imbalanced: active samples cluster on the body.
The reference has no equivalent (its only parallelism is DDP training, trainer.py:21-26).
"""
import torch
import torch.distributed as dist

DEFAULT_TILE = 512       # rays per tile (one image-row-sized strip of the compacted ray list)


def tile_indices(n_rays, rank, world, tile=DEFAULT_TILE, device='cpu'):
    """Ray indices owned by `rank`: tiles rank, rank+world, rank+2*world, ... of `tile` rays."""
    n_tiles = (n_rays + tile - 1) // tile
    mine = torch.arange(rank, n_tiles, world, device=device)
    idx = (mine[:, None] * tile + torch.arange(tile, device=device)[None, :]).reshape(-1)
    return idx[idx < n_rays]


def shard_counts(n_rays, world, tile=DEFAULT_TILE):
    return [int(tile_indices(n_rays, r, world, tile).numel()) for r in range(world)]


def gather_maps(local_rgba, n_rays, rank, world, tile=DEFAULT_TILE, group=None):
    """All-gather the per-rank [r,g,b,acc] rows (n_local,4) into the full (n_rays,4) map."""
    if world == 1:
        return local_rgba
    dev = local_rgba.device
    counts = shard_counts(n_rays, world, tile)
    mx = max(counts)
    send = torch.zeros(mx, 4, device=dev, dtype=local_rgba.dtype)
    send[:local_rgba.shape[0]] = local_rgba
    recv = torch.empty(world * mx, 4, device=dev, dtype=local_rgba.dtype)
    dist.all_gather_into_tensor(recv, send, group=group)
    full = torch.empty(n_rays, 4, device=dev, dtype=local_rgba.dtype)
    recv = recv.view(world, mx, 4)
    for r in range(world):
        full[tile_indices(n_rays, r, world, tile, device=dev)] = recv[r, :counts[r]]
    return full


def render_frame(render_fn, batch, rank, world, tile=DEFAULT_TILE, group=None):
    """render_fn(ray_o, ray_d, near, far) -> (rgb_map (n,3), acc_map (n,)) on this rank's rays.
    Returns the full-frame (rgb_map (n_rays,3), acc_map (n_rays,)) on every rank."""
    ray_o, ray_d, near, far = batch['ray_o'][0], batch['ray_d'][0], batch['near'][0], batch['far'][0]
    n = ray_o.shape[0]
    idx = tile_indices(n, rank, world, tile, device=ray_o.device)
    rgb, acc = render_fn(ray_o[idx], ray_d[idx], near[idx], far[idx])
    full = gather_maps(torch.cat([rgb, acc[:, None]], 1), n, rank, world, tile, group)
    return full[:, :3], full[:, 3]
