"""GPU: frames in flight (invr.frames.FrameSet) — K frames rendered as parallel branches of one hipGraph give, frame by frame, exactly
what K separate Renderer-level calls give (rgb / acc / raw / occ bit for bit), replay after replay, with per-frame workspaces sized
from a first render's survivor count; an undersized workspace is reported, never silently wrong."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(cfg_kw):
    import invr  # noqa: F401
    from invr import params
    from invr.config import make_cfg
    from invr.network import Network
    cfg = make_cfg(**cfg_kw)
    net = Network(cfg=cfg)
    net.load_state_dict(params.init_state_dict(cfg, seed=4), strict=True)
    return net.to('cuda:0').eval(), cfg


def test_frame_set_equals_separate_renders_and_replays_deterministically():
    from invr import scene, frames as iframes, dist as idist
    net, cfg = _net(dict(table_log2=12, N_samples=64))
    dev = torch.device('cuda', 0)
    batches = []
    for k in range(4):
        b, _ = scene.make_scene(128, 128, seed=0, cam_dist=1.8, frame=3 + 7 * k, pose_seed=k)
        batches.append({kk: v.to(dev) for kk, v in scene.to_torch(b).items()})
    for world, rank in ((1, 0), (4, 1)):                       # the whole frames, and rank 1's shard of a 4-way split
        refs = []
        for b in batches:
            idx = idist.tile_indices(b['ray_o'].shape[1], rank, world, device=dev)
            net._ws = None
            o = net.render_rays(b, b['ray_o'][0][idx], b['ray_d'][0][idx], b['near'][0][idx], b['far'][0][idx], 64, want_raw=True)
            refs.append({k: o[k].clone() for k in ('rgb_map', 'acc_map', 'raw', 'occ')})
        fns, n_rays, keep = iframes.shard_render_fns(net, batches, 64, rank, world, want_raw=True)
        fs = iframes.FrameSet(fns, n_rays, rank=0, world=1, device=dev)      # (no exchange: a group of one; the RCCL form is in test_gpu_rccl_world1.py)
        assert fs.graph is not None and fs.K == 4
        def mismatches():
            return [(k, key) for k in range(4) for key in ('rgb_map', 'acc_map', 'raw', 'occ') if not torch.equal(fs.local[k][key], refs[k][key])]
        for rep in range(3):
            fs.replay()
            torch.cuda.synchronize()
            assert iframes.check_overflow(fs)
            bad = mismatches()
            # hard again (round 6): the one-off mismatch round 5 turned into a warning was real — compiler-packed fp32 math in k_warp_pairs
            # returned wrong lanes beside the waves of other frames' kernels (profiles/r6_replay_mismatch.md); the library is built
            # without such instructions now and tests/test_gpu_frames.py::test_frames_in_flight_stress_* pins it at the bench shape
            assert not bad, (world, rep, [(k, key, int((fs.local[k][key] != refs[k][key]).sum()), float((fs.local[k][key] - refs[k][key]).abs().max())) for k, key in bad])
            for k in range(4):
                assert torch.equal(fs.full[k][:, :3], refs[k]['rgb_map']) and torch.equal(fs.full[k][:, 3], refs[k]['acc_map'])
        assert refs[0]['rgb_map'].shape != refs[1]['rgb_map'].shape or not torch.equal(refs[0]['rgb_map'], refs[1]['rgb_map'])   # the poses do differ


def test_frame_set_reports_an_undersized_workspace():
    from invr import scene, frames as iframes
    net, cfg = _net(dict(table_log2=12, N_samples=32))
    dev = torch.device('cuda', 0)
    b, _ = scene.make_scene(96, 96, seed=0, cam_dist=1.8)
    b = {kk: v.to(dev) for kk, v in scene.to_torch(b).items()}
    a = tuple(b[k][0].contiguous() for k in ('ray_o', 'ray_d', 'near', 'far'))
    ctx = net.prepare(b)
    full = net.render_rays(ctx, *a, 32, want_raw=False)
    na = int(full['stats'][0])
    assert na > 2000

    def fn():
        return net.render_rays(ctx, *a, 32, want_raw=False, max_active=na // 2)
    fs = iframes.FrameSet([fn], [a[0].shape[0]], device=dev)
    fs.replay()
    torch.cuda.synchronize()
    assert not iframes.check_overflow(fs)


def _sequence(n, res, dev):
    from invr import scene
    out = []
    for k in range(n):
        b, _ = scene.make_scene(res, res, seed=0, cam_dist=1.8, frame=(3 + 7 * k) % 100, pose_seed=k)
        out.append(scene.to_torch(b))
    return out


def test_frame_set_streams_mode_equals_separate_renders():
    """FrameSet(capture=False, streams=True): K eager launch chains on K streams, no graph — frame by frame what separate renders
    give, replay after replay (the replays of one frame queue on its own stream)."""
    from invr import frames as iframes
    net, cfg = _net(dict(table_log2=12, N_samples=64))
    dev = torch.device('cuda', 0)
    batches = [{kk: v.to(dev) for kk, v in b.items()} for b in _sequence(4, 128, dev)]
    refs = []
    for b in batches:
        net._ws = None
        o = net.render_rays(b, b['ray_o'][0], b['ray_d'][0], b['near'][0], b['far'][0], 64, want_raw=True)
        refs.append({k: o[k].clone() for k in ('rgb_map', 'acc_map', 'raw', 'occ')})
    fns, n_rays, keep = iframes.shard_render_fns(net, batches, 64, 0, 1, want_raw=True)
    fs = iframes.FrameSet(fns, n_rays, device=dev, capture=False, streams=True)
    assert fs.graph is None and fs.use_streams
    for rep in range(4):
        fs.replay()
        fs.replay()                                    # two replays back to back, no join in between
        torch.cuda.synchronize()
        assert iframes.check_overflow(fs)
        for k in range(4):
            for key in ('rgb_map', 'acc_map', 'raw', 'occ'):
                assert torch.equal(fs.local[k][key], refs[k][key]), (rep, k, key)


def test_run_evaluate_in_flight_equals_one_frame_at_a_time():
    """driver.run_evaluate over a 10-pose sequence with 4 frames in flight (Renderer.in_flight lanes: a stream + workspace each,
    dicts that join their frame on first access) = the strictly sequential loop, bit for bit; every batch has its own volume
    dimensions and ray count (nothing static to capture)."""
    from invr import driver
    from invr.renderer import Renderer
    net, cfg = _net(dict(table_log2=12, N_samples=64))
    dev = torch.device('cuda', 0)
    seq = _sequence(10, 96, dev)
    assert len({tuple(b['pbw'].shape) for b in seq}) > 1 or len({b['ray_o'].shape[1] for b in seq}) > 1
    one = driver.run_evaluate(net, seq, device=dev, in_flight=1, keep_maps=True)
    for K in (4, 3):
        many = driver.run_evaluate(net, seq, device=dev, in_flight=K, keep_maps=True)
        assert many['psnr'] == one['psnr'] and many['mse'] == one['mse']
        for a, b in zip(one['rgb_map'], many['rgb_map']):
            assert torch.equal(a, b)
    # the dict of a frame in flight: host maps, lazy raw / occ, device-resident variant, and an undersized survivor bound is re-rendered
    r = Renderer(net)
    r.in_flight = 2
    b0 = {k: v.to(dev) for k, v in seq[0].items()}
    b1 = {k: v.to(dev) for k, v in seq[1].items()}
    with torch.no_grad():
        r1 = Renderer(net)
        ref0, ref1 = r1.render(dict(b0)), r1.render(dict(b1))
        ref0, ref1 = dict(ref0), dict(ref1)
        r._cap_hint = 70000                             # far too small for frame 0: stats[6] fires, the join renders it again at full capacity
        a0 = r.render(dict(b0))
        a1 = r.render(dict(b1))
        assert 'raw' in a0 and len(a0) == 4 and set(a0.pending()) == {'rgb_map', 'acc_map', 'raw', 'occ'}
        for got, ref in ((a1, ref1), (a0, ref0)):
            for k in ('rgb_map', 'acc_map', 'raw', 'occ'):
                assert not got[k].is_cuda and torch.equal(got[k], ref[k]), k
        r.eval_to_cpu = False
        d0, d1, d2 = r.render(dict(b0)), r.render(dict(b1)), r.render(dict(b0))          # the third call joins the first (lane reuse)
        assert not d0.in_flight()
        for got, ref in ((d2, ref0), (d1, ref1), (d0, ref0)):
            for k in ('rgb_map', 'acc_map', 'raw', 'occ'):
                assert got[k].is_cuda and torch.equal(got[k].cpu(), ref[k]), k
        r.flush()


def _bench_frames(K):
    """the bench's model and its K-frame sequence (bench.build_model / bench.frame_batches: 512 x 512 x 128, the full 1.09 GB tables)"""
    import bench
    from invr.config import make_cfg
    dev = torch.device('cuda', 0)
    cfg = make_cfg(N_samples=128)
    cfg['eval_row_sums'] = True
    net = bench.build_model(cfg, dev)
    _, batches = bench.frame_batches(512, 1.8, K, dev)
    refs = []
    for b in batches:
        net._ws = None
        o = net.render_rays(b, b['ray_o'][0], b['ray_d'][0], b['near'][0], b['far'][0], 128, want_raw=True)
        refs.append({k: o[k].clone() for k in ('rgb_map', 'acc_map', 'raw')})
    net._ws = None
    return net, batches, refs, dev


def _first_difference(got, ref, what):
    ne = (got.reshape(ref.shape).view(torch.int32) != ref.view(torch.int32))
    idx = ne.reshape(-1).nonzero()[:4, 0].tolist()
    return '%s: %d differing elements, first at %r: got %r, expected %r' % (
        what, int(ne.sum()), idx, [float(got.reshape(-1)[i]) for i in idx], [float(ref.reshape(-1)[i]) for i in idx])


def test_frames_in_flight_stress_graph_at_the_bench_shape():
    """the mode the headline is timed in — 10 frames of the 512 x 512 x 128 sequence as branches of one hipGraph, full
    tables — replayed 250 times; every replay, every frame, rgb_map / acc_map / raw bit for bit what a separate render gives.  Fails on
    the first differing element.  (Round 5's library failed this in ~1 of 30 replays; tools/stress_replay.py is the long form and names
    the kernel stage that diverged.)"""
    from invr import frames as iframes
    net, batches, refs, dev = _bench_frames(10)
    fns, n_rays, keep = iframes.shard_render_fns(net, batches, 128, 0, 1, want_raw=True)
    fs = iframes.FrameSet(fns, n_rays, device=dev)
    assert fs.graph is not None and fs.K == 10
    for rep in range(250):
        fs.replay()
        torch.cuda.synchronize()
        for k in range(10):
            for key in ('rgb_map', 'acc_map', 'raw'):
                if not torch.equal(fs.local[k][key], refs[k][key]):
                    pytest.fail('replay %d, frame %d, %s' % (rep, k, _first_difference(fs.local[k][key], refs[k][key], key)))
    assert iframes.check_overflow(fs)


def test_frames_in_flight_stress_renderer_lanes_at_the_bench_shape():
    """... and the drop-in call: Renderer.render with in_flight = 8 (eager launch chains on 8 streams, workspaces reused by the frames
    that land on a lane) over 40 passes of the 10-frame sequence, device outputs, read 7 calls later as driver.run_evaluate does."""
    from collections import deque
    from invr.renderer import Renderer
    net, batches, refs, dev = _bench_frames(10)
    r = Renderer(net)
    r.in_flight, r.eval_to_cpu = 8, False
    q = deque()

    def check(o, k, n):
        for key in ('rgb_map', 'acc_map', 'raw'):
            got = o[key][0]
            if not torch.equal(got, refs[k][key]):
                pytest.fail('frame %d of the run (frame %d of the sequence), %s' % (n, k, _first_difference(got, refs[k][key], key)))
    for n in range(400):
        k = n % 10
        q.append((r.render(dict(batches[k])), k, n))
        while len(q) >= r.in_flight:
            check(*q.popleft())
    while q:
        check(*q.popleft())
    r.flush(release=True)
