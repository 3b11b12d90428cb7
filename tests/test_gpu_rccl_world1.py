"""GPU: the RCCL calls of the multi-GPU paths, executed for real on a 1-GPU box.

RCCL refuses two ranks on one device, so the two-rank tests use gloo and the multi-GPU bench has never run on this hardware
pool ("RCCL paths have never executed anywhere").  A process group of ONE rank on backend "nccl" (= RCCL on ROCm)
still runs every collective through librccl: with INVR_FORCE_COLLECTIVES=1 the world-size-1 shortcuts are off and this test
executes, on cuda:0,
  * dist.gather_maps: the padded device-tensor all_gather_into_tensor + index_select of the eval tiles, inside bench.py's own loop
    shape (hipGraph replay of the frame, then the collective on the same stream),
  * dist_train.GradReducer: ReduceOp.AVG all-reduces of the row-scalar table gradients and of the flat small-tensor buffer with
    async handles, started from inside TrainRenderFn.backward between the part chains and the deformer stage, joined by the
    optimiser's pre-hook — through NetworkWrapper + FusedAdam for 3 steps,
  * frames.FrameSet: three frames as branches of one hipGraph with ONE all_gather_into_tensor + the index_selects captured in the graph,
and checks that a group of one leaves the numbers untouched (the collectives are identities there)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), INVR_FORCE_COLLECTIVES='1', RANK='0', WORLD_SIZE='1')
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import invr  # noqa: F401
    from invr import scene, params, driver, dist as idist, dist_train
    from invr.config import make_cfg
    from invr.network import Network
    from invr.trainer import NetworkWrapper
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    res = {}
    try:
        assert dist.get_backend() == 'nccl'
        cfg = make_cfg(table_log2=12, N_samples=32)
        sd = params.init_state_dict(cfg, seed=5)
        net = Network(cfg=cfg)
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).eval()
        bnp, _ = scene.make_scene(96, 96, seed=1, cam_dist=1.8)
        gb = {k: v.to(dev) for k, v in scene.to_torch(bnp).items()}
        ro, rd, nr, fa = (gb[k][0] for k in ('ray_o', 'ray_d', 'near', 'far'))
        n = ro.shape[0]
        ctx = net.prepare(gb)

        def render():
            o = net.render_rays(ctx, ro, rd, nr, fa, 32, want_raw=False)
            return torch.cat([o['rgb_map'], o['acc_map'][:, None]], 1)
        ref = render().clone()
        # eval: graph replay + all-gather, as bench.py's step
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            render()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            rgba = render()
        for _ in range(3):
            g.replay()
            full = idist.gather_maps(rgba, n, 0, 1)
        torch.cuda.synchronize()
        res['gather_equal'] = bool(torch.equal(full, ref))
        torch.save(res, out_path)
        res['gather_shape'] = tuple(full.shape)
        # the frame server's form (bench.py --gpus N > 1): ONE exchange in flight, joined after the next frame has been replayed and sent
        pend, fulls = None, []
        for _ in range(4):
            g.replay()
            nxt = idist.gather_maps_async(rgba.clone(), n, 0, 1)
            if pend is not None:
                fulls.append(pend.result())
            pend = nxt
        fulls.append(pend.result())
        torch.cuda.synchronize()
        res['async_gather_equal'] = len(fulls) == 4 and all(bool(torch.equal(f, ref)) for f in fulls)
        torch.save(res, out_path)
        # frames in flight (bench.py --gpus N): K frames as parallel branches of ONE hipGraph with the RCCL all-gather of their tiles and the
        # index_selects CAPTURED in the same graph — one replay per K frames, nothing issued from the host per frame
        from invr import frames as iframes
        fbatches = []
        for k in range(3):
            fb, _ = scene.make_scene(96, 96, seed=1, cam_dist=1.8, frame=5 + k, pose_seed=k)
            fbatches.append({kk: v.to(dev) for kk, v in scene.to_torch(fb).items()})
        fns, nrs, keep = iframes.shard_render_fns(net, fbatches, 32, 0, 1, want_raw=False)
        singles = [iframes.FrameSet._rgba(fn()).clone() for fn in fns]
        fs = iframes.FrameSet(fns, nrs, rank=0, world=1, device=dev)
        res['frameset_exchange_captured'] = bool(fs.exchange and fs.graph is not None and fs.exchange_captured)
        for _ in range(3):
            fs.replay()
        torch.cuda.synchronize()
        res['frameset_equal'] = all(bool(torch.equal(f, s_)) for f, s_ in zip(fs.full, singles)) and iframes.check_overflow(fs) and fs.own_rows_match()
        res['frameset_poses_differ'] = singles[0].shape != singles[1].shape or not bool(torch.equal(singles[0], singles[1]))
        torch.save(res, out_path)
        # training: averaged gradients through the reducer (AVG over one rank = identity)
        finals = []
        for use_reducer in (True, False):
            net2 = Network(cfg=cfg)
            net2.load_state_dict(sd, strict=True)
            net2 = net2.to(dev).train()
            wrap = NetworkWrapper(net2)
            gen = torch.Generator().manual_seed(9)
            bp, _ = scene.make_scene(512, 512, seed=2, frame=11, cam_dist=1.8, crop=(240, 252, 20, 20))
            tb = {k: v.to(dev) for k, v in scene.to_torch(bp).items()}
            nr_ = tb['ray_o'].shape[1]
            jit = torch.rand(nr_, 32, generator=gen).to(dev)
            noi = torch.rand(nr_ * 32 * 5, 3, generator=gen).to(dev)
            wrap.renderer._jitter = lambda shape, device: jit
            wrap.renderer._pair_noise_dense = lambda rows, device: noi[:rows]
            opt = driver.make_optimizer(net2, lr=1e-3, eps=1e-15)
            if use_reducer:
                red = dist_train.attach(opt)
                dist_train.broadcast_parameters(net2)
            for it in range(3):
                loss, _ = driver.train_step(wrap, opt, dict(tb), it + 2)
            torch.cuda.synchronize()
            finals.append({k: v.detach().cpu() for k, v in net2.state_dict().items()})
            if use_reducer:
                res['reducer_pending_after_step'] = len(red.pending)
        worst = 0.0
        for k in finals[0]:
            if finals[0][k].is_floating_point():
                d = (finals[0][k].double() - finals[1][k].double()).abs()
                worst = max(worst, float(d.max()))
                frac = float((d <= 1e-6 + 1e-5 * finals[1][k].double().abs()).double().mean())
                assert frac >= 0.99, (k, frac)
        res['train_worst_abs_diff'] = worst
        res['loss'] = float(loss)
        # the reference's own distributed form (trainer.py:18-26): DistributedDataParallel(NetworkWrapper) + ITS optimizer (torch.optim.Adam,
        # one group per tensor) around the FUSED training node — the node returns ordinary dense gradients when no arena is attached, so
        # DDP's bucket hooks see every parameter; a group of one must leave the plain (non-DDP) run's parameters untouched
        from torch.nn.parallel import DistributedDataParallel as DDP
        ddp_final = []
        for use_ddp in (True, False):
            net3 = Network(cfg=cfg)
            net3.load_state_dict(sd, strict=True)
            net3 = net3.to(dev).train()
            wrap3 = NetworkWrapper(net3)
            wrap3.renderer._jitter = lambda shape, device: jit
            wrap3.renderer._pair_noise_dense = lambda rows, device: noi[:rows]
            model = DDP(wrap3, device_ids=[0]) if use_ddp else wrap3
            opt3 = driver.make_optimizer(net3, lr=1e-3, eps=1e-15, fused=False)
            assert type(opt3) is torch.optim.Adam and len(opt3.param_groups) == sum(1 for p in net3.parameters() if p.requires_grad)
            for it in range(2):
                b3 = dict(tb)
                b3['iter_step'] = it + 2
                ret3, loss3, _, _ = model(b3, 0, split='train')
                loss3 = loss3.mean()
                opt3.zero_grad(set_to_none=True)
                loss3.backward()
                opt3.step()
            torch.cuda.synchronize()
            ddp_final.append({k: v.detach().cpu() for k, v in net3.state_dict().items()})
        # (same closeness rule as above: the backward's float atomics are unordered, and Adam with eps 1e-15 turns a rounding-level
        # gradient difference into a +-lr step on a few elements)
        ddp_worst, ddp_frac = 0.0, 1.0
        for k in ddp_final[0]:
            if ddp_final[0][k].is_floating_point():
                d = (ddp_final[0][k].double() - ddp_final[1][k].double()).abs()
                ddp_worst = max(ddp_worst, float(d.max()))
                ddp_frac = min(ddp_frac, float((d <= 1e-6 + 1e-5 * ddp_final[1][k].double().abs()).double().mean()))
        res['ddp_worst'], res['ddp_frac'] = ddp_worst, ddp_frac
        res['ddp_equal'] = ddp_frac >= 0.99 and ddp_worst <= 2 * 2 * 1e-3 * 1.01
        res['ddp_loss'] = float(loss3)
        res['ok'] = True
    finally:
        torch.save(res, out_path)
        dist.destroy_process_group()


def test_rccl_collectives_execute_on_one_gpu(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'res.pt')
    ctx = mp.get_context('spawn')
    p = ctx.Process(target=_worker, args=(port, out))
    p.start()
    p.join(600)
    res = torch.load(out) if os.path.exists(out) else {}
    assert p.exitcode == 0, (p.exitcode, res)          # (the worker saves its partial results: a crash names the stage it got to)
    assert res.get('ok'), res
    assert res['gather_equal'] and res['gather_shape'][1] == 4
    assert res['async_gather_equal']
    assert res['frameset_exchange_captured'] and res['frameset_equal'] and res['frameset_poses_differ']
    assert res['reducer_pending_after_step'] == 0                      # the optimiser pre-hook joined every async all-reduce
    assert res['train_worst_abs_diff'] <= 2 * 3 * 1e-3 * 1.01 and res['loss'] == res['loss']
    assert res['ddp_equal'] and res['ddp_loss'] == res['ddp_loss']      # DistributedDataParallel(NetworkWrapper) around the fused node
