"""Multi-rank path on CPU: world_size-2 gloo.  The tile-cyclic ray sharding and the single
all-gather of [r,g,b,acc] tiles (invr/dist.py) must reproduce the single-rank frame exactly.
The per-rank renderer is a deterministic stand-in (a pure function of the rays): the exchange
logic is what is under test here, the HIP renderer itself is covered by the -m gpu tests."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from invr import dist as idist


def fake_render(ray_o, ray_d, near, far):
    rgb = torch.stack([torch.sin(ray_d[:, 0] * 3 + near), torch.cos(ray_d[:, 1] * 5 + far), ray_o[:, 2] + ray_d[:, 2]], 1)
    return rgb, (near * far).sin()


def make_batch(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {'ray_o': torch.rand(1, n, 3, generator=g), 'ray_d': torch.rand(1, n, 3, generator=g),
            'near': torch.rand(1, n, generator=g), 'far': torch.rand(1, n, generator=g) + 1}


def _worker(rank, world, port, n, tile, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        batch = make_batch(n)
        rgb, acc = idist.render_frame(fake_render, batch, rank, world, tile=tile)
        ref_rgb, ref_acc = fake_render(batch['ray_o'][0], batch['ray_d'][0], batch['near'][0], batch['far'][0])
        ok = torch.equal(rgb, ref_rgb) and torch.equal(acc, ref_acc)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, n, tile):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, tile, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_tile_partition_is_exact_cover():
    for n, world, tile in ((1, 2, 512), (1000, 2, 64), (4097, 8, 512), (262144, 8, 512), (5, 4, 2)):
        idx = torch.cat([idist.tile_indices(n, r, world, tile) for r in range(world)])
        assert idx.numel() == n and torch.equal(idx.sort()[0], torch.arange(n))
        counts = idist.shard_counts(n, world, tile)
        assert sum(counts) == n and max(counts) - min(counts) <= tile


def _async_worker(rank, world, port, n, tile, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # a frame server with ONE gather in flight: frame f's exchange is joined after frame f+1 has been rendered and sent
        frames = [make_batch(n, seed=s) for s in range(4)]
        idx = idist.tile_indices(n, rank, world, tile)
        pending, got = None, []
        for b in frames:
            rgb, acc = fake_render(b['ray_o'][0][idx], b['ray_d'][0][idx], b['near'][0][idx], b['far'][0][idx])
            nxt = idist.gather_maps_async(torch.cat([rgb, acc[:, None]], 1), n, rank, world, tile)
            if pending is not None:
                got.append(pending.result())
            pending = nxt
        got.append(pending.result())
        ok = True
        for b, full in zip(frames, got):
            ref_rgb, ref_acc = fake_render(b['ray_o'][0], b['ray_d'][0], b['near'][0], b['far'][0])
            ok = ok and torch.equal(full[:, :3], ref_rgb) and torch.equal(full[:, 3], ref_acc)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_async_gather_world2_one_frame_in_flight():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_async_worker, args=(2, _free_port(), 1000, 64, ret), nprocs=2, join=True)
    assert all(ret[r] for r in range(2)), dict(ret)


def _frameset_worker(rank, world, port, sizes, tile, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # K frames of different sizes per exchange (invr.frames.FrameSet, eager on CPU: the graph capture is a GPU matter): ONE
        # all-gather for all of them, every rank ends with every frame's full map in ray order
        from invr import frames as iframes
        frames = [make_batch(n, seed=10 + k) for k, n in enumerate(sizes)]
        fns = []
        for b, n in zip(frames, sizes):
            idx = idist.tile_indices(n, rank, world, tile)

            def fn(b=b, idx=idx):
                rgb, acc = fake_render(b['ray_o'][0][idx], b['ray_d'][0][idx], b['near'][0][idx], b['far'][0][idx])
                return {'rgb_map': rgb, 'acc_map': acc}
            fns.append(fn)
        fs = iframes.FrameSet(fns, sizes, rank=rank, world=world, device='cpu', tile=tile, capture=False)
        ok = fs.exchange and fs.plan['rows'] == sum(max(idist.shard_counts(n, world, tile)) for n in sizes)
        for _ in range(2):                                           # a second replay reuses the buffers
            fs.replay()
            ok = ok and fs.own_rows_match()
            for b, full in zip(frames, fs.full):
                ref_rgb, ref_acc = fake_render(b['ray_o'][0], b['ray_d'][0], b['near'][0], b['far'][0])
                ok = ok and torch.equal(full[:, :3], ref_rgb) and torch.equal(full[:, 3], ref_acc)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_frame_set_world2_ragged_frames_one_exchange():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_frameset_worker, args=(2, _free_port(), (1000, 30, 517, 64), 64, ret), nprocs=2, join=True)
    assert all(ret[r] for r in range(2)), dict(ret)


def test_frame_set_single_rank_is_the_local_render():
    from invr import frames as iframes
    b = make_batch(300, seed=3)
    rgb, acc = fake_render(b['ray_o'][0], b['ray_d'][0], b['near'][0], b['far'][0])
    fs = iframes.FrameSet([lambda: torch.cat([rgb, acc[:, None]], 1)], [300], device='cpu', capture=False)
    fs.replay()
    assert not fs.exchange and torch.equal(fs.full[0][:, :3], rgb) and torch.equal(fs.full[0][:, 3], acc)


def test_gather_world2_ragged():
    _run(2, 1000, 64)          # 16 tiles, last one ragged (40 rays)


def test_gather_world2_fewer_tiles_than_ranks():
    _run(2, 30, 64)            # one tile only: rank 1 renders nothing


# ---- data-parallel training: the gradient reducer (invr/dist_train.py) on CPU stand-ins ---------------------------------
class _FakeEmbedder:
    def __init__(self, n, seed):
        self._rg = torch.randn(n, generator=torch.Generator().manual_seed(seed))

    def row_grad(self):
        return self._rg


class _FakeArena:
    def __init__(self, rank):
        self.embedders = [_FakeEmbedder(n, 100 * rank + k) for k, n in enumerate((1000, 50, 7000, 3, 3))]
        self.flat = torch.randn(123, generator=torch.Generator().manual_seed(7 + rank))


def _reduce_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from invr import dist_train
        arenas = [_FakeArena(r) for r in range(world)]          # every rank can compute the expected mean locally
        mine = arenas[rank]
        red = dist_train.GradReducer(mine)
        assert red.part_order == [2, 0, 1, 3, 4]                # largest block first
        for p in red.part_order:
            red.reduce_part(p)
        red.reduce_small()
        red.wait()
        ok = True
        for k in range(5):
            want = torch.stack([_FakeArena(r).embedders[k].row_grad() for r in range(world)]).mean(0)
            ok = ok and torch.allclose(mine.embedders[k].row_grad(), want, atol=1e-7)
        want = torch.stack([_FakeArena(r).flat for r in range(world)]).mean(0)
        ok = ok and torch.allclose(mine.flat, want, atol=1e-7)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_averages_every_block():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reduce_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert all(ret[r] for r in range(2)), dict(ret)


# ---- world 8: the node size of BASELINE configs[2] / configs[4] (one process per GPU; 8 gloo ranks on the CPU here) ----------
def test_frame_set_world8_ragged_frames_one_exchange():
    """8 ranks, K = 5 frames of different sizes — incl. a frame with fewer tiles than ranks (ranks that render nothing) and a
    ragged last tile: ONE padded all-gather per replay, every rank ends with every full frame, replay after replay."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_frameset_worker, args=(8, _free_port(), (4097, 30, 517, 64, 1000), 64, ret), nprocs=8, join=True)
    assert all(ret[r] for r in range(8)), dict(ret)


def test_gather_world8_ragged_and_fewer_tiles_than_ranks():
    _run(8, 4097, 64)          # 65 tiles over 8 ranks: shards of 9 / 8 tiles, last tile one ray
    _run(8, 200, 64)           # 4 tiles: ranks 4..7 render nothing


def test_async_gather_world8_one_frame_in_flight():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_async_worker, args=(8, _free_port(), 3000, 64, ret), nprocs=8, join=True)
    assert all(ret[r] for r in range(8)), dict(ret)


def test_grad_reducer_world8_averages_every_block():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reduce_worker, args=(8, _free_port(), ret), nprocs=8, join=True)
    assert all(ret[r] for r in range(8)), dict(ret)


def test_tile_indices_rank_beyond_the_tile_count_is_empty():
    """(the world-8 FrameSet test's bug: torch.arange(rank, n_tiles, world) raises for rank > n_tiles)"""
    for n, world, tile in ((30, 8, 64), (1, 8, 512), (130, 8, 64), (0, 4, 64)):
        got = [idist.tile_indices(n, r, world, tile) for r in range(world)]
        assert sorted(torch.cat(got).tolist()) == list(range(n))
        assert all(g.dtype == torch.int64 for g in got)
