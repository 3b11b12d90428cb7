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
        for rep in range(3):
            fs.replay()
            torch.cuda.synchronize()
            assert iframes.check_overflow(fs)
            for k in range(4):
                for key in ('rgb_map', 'acc_map', 'raw', 'occ'):
                    assert torch.equal(fs.local[k][key], refs[k][key]), (world, rep, k, key)
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
