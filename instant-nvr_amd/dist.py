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


def FORCE_COLLECTIVES():
    """INVR_FORCE_COLLECTIVES=1: issue the collectives even in a group of one rank — lets a 1-GPU box execute the very RCCL calls
    (device all_gather_into_tensor, ReduceOp.AVG all-reduce, async handles) the multi-GPU runs make (tests/test_gpu_rccl_world1.py)."""
    import os
    return os.environ.get('INVR_FORCE_COLLECTIVES', '0') == '1'


def tile_indices(n_rays, rank, world, tile=DEFAULT_TILE, device='cpu'):
    """Ray indices owned by `rank`: tiles rank, rank+world, rank+2*world, ... of `tile` rays."""
    n_tiles = (n_rays + tile - 1) // tile
    if rank >= n_tiles:              # fewer tiles than ranks: this rank renders nothing (torch.arange refuses start > end)
        return torch.empty(0, dtype=torch.int64, device=device)
    mine = torch.arange(rank, n_tiles, world, device=device)
    idx = (mine[:, None] * tile + torch.arange(tile, device=device)[None, :]).reshape(-1)
    return idx[idx < n_rays]


def shard_counts(n_rays, world, tile=DEFAULT_TILE):
    return [int(tile_indices(n_rays, r, world, tile).numel()) for r in range(world)]


_plan_cache = {}


def gather_plan(n_rays, world, tile, device):
    """(max shard size, src_index): ray i of the full frame sits at row src_index[i] of the gathered
    (world*mx, 4) buffer.  Cached: the plan only depends on the frame size and the world size."""
    key = (n_rays, world, tile, str(device))
    if key not in _plan_cache:
        counts = shard_counts(n_rays, world, tile)
        mx = max(counts)
        src = torch.empty(n_rays, dtype=torch.int64)
        for r in range(world):
            idx = tile_indices(n_rays, r, world, tile)
            src[idx] = r * mx + torch.arange(idx.numel())
        _plan_cache[key] = (mx, src.to(device))
    return _plan_cache[key]


def gather_maps(local_rgba, n_rays, rank, world, tile=DEFAULT_TILE, group=None):
    """All-gather the per-rank [r,g,b,acc] rows (n_local,4) into the full (n_rays,4) map:
    one padded all_gather_into_tensor + one index_select."""
    if world == 1 and not (FORCE_COLLECTIVES() and dist.is_initialized()):
        return local_rgba
    dev = local_rgba.device
    mx, src = gather_plan(n_rays, world, tile, dev)
    if local_rgba.shape[0] == mx:
        send = local_rgba.contiguous()
    else:
        send = torch.zeros(mx, 4, device=dev, dtype=local_rgba.dtype)
        send[:local_rgba.shape[0]] = local_rgba
    recv = torch.empty(world * mx, 4, device=dev, dtype=local_rgba.dtype)
    if dist.get_backend(group) == 'gloo' and send.is_cuda:      # gloo (tests) gathers through host memory
        host = torch.empty(world * mx, 4, dtype=local_rgba.dtype)
        dist.all_gather_into_tensor(host, send.cpu(), group=group)
        recv.copy_(host)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    return recv.index_select(0, src)


class PendingFrame:
    """An all-gather in flight (gather_maps_async): `result()` joins it and returns the full (n_rays, 4) map."""

    def __init__(self, work, recv, src, keep):
        self.work, self.recv, self.src, self.keep = work, recv, src, keep

    def result(self):
        if self.work is not None:
            self.work.wait()                       # nccl: orders the current stream behind the collective (no host synchronisation)
            self.work = None
        return self.recv if self.src is None else self.recv.index_select(0, self.src)


def gather_maps_async(local_rgba, n_rays, rank, world, tile=DEFAULT_TILE, group=None):
    """gather_maps with the collective left in flight: the caller renders the next frame beside it and calls `.result()` when it needs
    the full map (a frame server keeps one gather pending: the 4 MB exchange of frame f overlaps the kernels of frame f+1).  The send
    buffer is kept alive by the returned object."""
    if world == 1 and not (FORCE_COLLECTIVES() and dist.is_initialized()):
        return PendingFrame(None, local_rgba, None, None)
    dev = local_rgba.device
    mx, src = gather_plan(n_rays, world, tile, dev)
    if local_rgba.shape[0] == mx:
        send = local_rgba.clone()              # a private buffer: the caller may overwrite its rows (a graph's output) before result()
    else:
        send = torch.zeros(mx, 4, device=dev, dtype=local_rgba.dtype)
        send[:local_rgba.shape[0]] = local_rgba
    if dist.get_backend(group) == 'gloo' and send.is_cuda:      # gloo (tests) gathers through host memory: synchronous
        return PendingFrame(None, gather_maps(local_rgba, n_rays, rank, world, tile, group), None, None)
    recv = torch.empty(world * mx, 4, device=dev, dtype=local_rgba.dtype)
    work = dist.all_gather_into_tensor(recv, send, group=group, async_op=True)
    return PendingFrame(work, recv, src, send)


def render_frame(render_fn, batch, rank, world, tile=DEFAULT_TILE, group=None):
    """render_fn(ray_o, ray_d, near, far) -> (rgb_map (n,3), acc_map (n,)) on this rank's rays.
    Returns the full-frame (rgb_map (n_rays,3), acc_map (n_rays,)) on every rank."""
    ray_o, ray_d, near, far = batch['ray_o'][0], batch['ray_d'][0], batch['near'][0], batch['far'][0]
    n = ray_o.shape[0]
    idx = tile_indices(n, rank, world, tile, device=ray_o.device)
    rgb, acc = render_fn(ray_o[idx], ray_d[idx], near[idx], far[idx])
    full = gather_maps(torch.cat([rgb, acc[:, None]], 1), n, rank, world, tile, group)
    return full[:, :3], full[:, 3]
