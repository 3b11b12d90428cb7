#!/usr/bin/env python
"""Bench of the render hot path (BASELINE.json metric: ray-samples/s).

One step = one 512x512 frame x N_samples=128 of the synthetic ZJU-377-shaped scene rendered by
Renderer-level code through libinvr.so with the full-size inb_377 model (285,993,711 parameters,
1.09 GB of hash tables, random init N(0,0.1^2) tables — there is no dataset / checkpoint here).
Inputs (rays, scene tensors, parameters) are resident in HBM before the timed region; the step ends
with rgb_map/acc_map (and the reference's raw/occ outputs) in HBM.

Frames in flight: the timed frames are K (--in-flight, default 10; a divisor of --steps) frames of a synthetic sequence — the same body
in K poses — rendered side by side by ONE hipGraph replay (invr.frames.FrameSet: K parallel branches of one captured graph).  A step is
one frame; every frame does all of its per-frame work.

N GPUs (python -m torch.distributed.run ... bench.py --gpus N): every frame's rays are dealt to the ranks tile-cyclically, every rank
renders its tiles of the K frames with a full model replica, and ONE RCCL all-gather of the K frames' [r,g,b,acc] tiles — captured in
the same graph — assembles them on every rank; total work is fixed -> "scaling": "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      — the dominant roofline-bound stage: the part MLPs (k_part_occ_all + k_winner_lists + k_part_rgb_all): algorithmic
                  FLOPs = 4.6 k x listed pairs + 17.5 k / 9.3 k x winning pairs (SURVEY.md §8d's 22.1 k / 14.0 k per pair split into
                  its two MLPs) / the stage's HIP-event time, vs the 157.3 TFLOP/s of fp32-in MFMA; measured HBM bytes and busy
                  counters of the same command from profiles/ (PMC passes cannot run inside the timed region; `counters_stale` says
                  whether they were measured on the kernel sources this line ran)
  roofline_other— the KNN (VALU-issue bound, no roofline fraction) and the encoder (measured bytes primary, the byte model secondary)
  path_roofline — the WHOLE frame: measured HBM / Infinity-Cache bytes per frame primary, SURVEY §8d's byte model labelled secondary
  shard_projection, samples_64, mid_density, aggr_mean, full_rows, dense_stress, api_frame — variants of the headline frames (N=1 only): rank 0's
                  shard of a W-way split (W = 1, 2, 4, 8) on this one GPU with the headline's frames in flight and with one frame at a
                  time, the yaml-default 64 samples/ray, smpl_thresh 0.1 (3x the survivors), the 64-byte-row encoder, the dense
                  stress frame (every sample survives), and the wall clock of Renderer.render(batch) with and without the reference's
                  move-everything-to-the-host contract
  cpu_baseline  — the oracle (CPU PyTorch port of the reference path) timed on the host cores on a
                  bounded sample of the same workload (N=1 only): one 4096-ray chunk + BASELINE configs[0]
  train_step, api_train_step — informational: configs[4]-shaped training iterations with the build's fused optimizer and with the
                  optimizer exactly as the reference's train_net.py builds it (N=1 only)
The timed region is blocks of exactly --steps frames between fences, repeated until >= --min-time seconds.

`--train [--train-config 377|lan] [--gpus N]` benches the training iteration instead (BASELINE configs[4] / configs[3]:
fused forward + backward + dense Adam over the full-size model, data-parallel over N ranks), see main_train.
`--shard-of W` renders rank 0's shard of a W-way split on one GPU (strong-scaling evidence without an 8-GPU node).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch                               # noqa: E402
import torch.distributed as dist           # noqa: E402

import invr                                # noqa: E402,F401
from invr import scene as scene_mod, _abi  # noqa: E402
from invr import dist as idist             # noqa: E402
from invr import frames as iframes         # noqa: E402
from invr.config import make_cfg           # noqa: E402
from invr.network import Network           # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)
ROW_BYTES = 64             # one 16-feature fp32 table row
PAIR_TABLE_BYTES = 16 * 8 * ROW_BYTES      # 16 levels x 8 corners x 64 B = 8192 B per (point,part) pair


def csrc_digest():
    """sha256 over the kernel sources the library is built from: the PMC summaries under profiles/ record it (tools/prof_all.sh), and a
    bench line that reads counters measured on OTHER sources says so (`counters_stale`)."""
    import hashlib
    d = os.path.join(ROOT, 'instant-nvr_amd', 'csrc')
    h = hashlib.sha256()
    for n in sorted(os.listdir(d)):
        if n.endswith(('.hip', '.h')):
            h.update(n.encode())
            h.update(open(os.path.join(d, n), 'rb').read())
    return h.hexdigest()[:16]


def build_model(cfg, device, seed=0):
    # the MLPs take torch's default initialisation: seeded, because WHICH part wins a survivor's max-occupancy merge — and with it how
    # many pairs run the (larger) body / head colour MLP — depends on those weights; unseeded, the colour kernel's time moved between
    # 0.25 and 0.43 ms from run to run (round 3 first read that as a power-state effect)
    torch.manual_seed(seed)
    with torch.device(device):
        net = Network(cfg=cfg)
    net = net.to(device).eval()
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith('embedder.dense') or name.endswith('embedder.hash'):
                p.normal_(0.0, 0.1, generator=g)
    return net


def cpu_baseline(net, cfg, batch_cpu, n_rays, S, seed=0):
    """The oracle (kind "port": this repo's CPU PyTorch restatement of the reference path, validated against the imported
    reference on the golden fixtures) on the host cores, as BASELINE.md §3 plans it: one 4096-ray x S chunk of the
    bench frame (`value`) and BASELINE configs[0] (C1: a 64x64 frame x 32 samples), warm, with the torch thread count
    picked by a quick sweep on a 512-ray sample (every intra-op fork over all 256 hardware threads of the GPU box made
    the round-1 figure 100x too low)."""
    from oracle import nvr_oracle as O     # checker / baseline only — never on the product path
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    model = O.Model(sd, cfg)
    n = batch_cpu['ray_o'].shape[1]
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed))

    def sub(batch, sel):
        b = dict(batch)
        for k in ('ray_o', 'ray_d', 'near', 'far'):
            b[k] = batch[k][:, sel]
        return b

    def timed(b, s, chunk):
        with torch.no_grad():
            t0 = time.time()
            O.render(model, b, n_samples=s, chunk=chunk)
            return time.time() - t0
    probe = sub(batch_cpu, perm[:512].sort()[0])
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (4, 8, 16, 32, 64, 128, ncpu) if t <= ncpu})
    sweep = {}
    for t in cands:
        torch.set_num_threads(t)
        timed(probe, S, 512)                     # warm (allocator, thread pool)
        sweep[t] = 512 * S / timed(probe, S, 512)
        if t >= 16 and sweep[t] < 0.5 * max(sweep.values()):
            break                                # past the knee: do not spend the budget on oversubscribed settings
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    chunk_rays = min(n_rays, n)
    b = sub(batch_cpu, perm[:chunk_rays].sort()[0])
    dt = timed(b, S, 4096)
    # C1 (BASELINE configs[0]): 64x64 frame x 32 samples of the same model
    c1_np, _ = scene_mod.make_scene(64, 64, seed=0)
    c1 = scene_mod.to_torch(c1_np)
    timed(c1, 32, 4096)
    dt1 = timed(c1, 32, 4096)
    n1 = c1['ray_o'].shape[1] * 32
    return {'value': chunk_rays * S / dt, 'unit': 'ray-samples/s', 'cores': best, 'kind': 'port',
            'sample': 'one %d-ray x %d-sample chunk of the bench frame (full 1.09 GB tables), oracle/nvr_oracle.py, torch %s CPU, '
                      '%d of %d host threads (best of sweep %s), %.1f s warm'
                      % (chunk_rays, S, torch.__version__, best, ncpu, {k: round(v) for k, v in sweep.items()}, dt),
            'c1': {'value': n1 / dt1, 'unit': 'ray-samples/s', 'sample': 'BASELINE configs[0]: %d rays (64x64 frame) x 32 samples, %.2f s warm'
                   % (c1['ray_o'].shape[1], dt1)}}


def train_probe(net, dev, S, iters, fused=None):
    """BASELINE configs[4]-shaped training iteration on this GPU: a 32x32 patch (1024 rays) x S samples, forward +
    backward (HIP kernels behind autograd) + the Adam step over all parameters; informational extra object.  fused=None: the
    build's own driver.make_optimizer (FusedAdam: one launch, row-scalar table gradients); fused=False: the optimizer exactly as
    the reference's train_net.py builds it (lib/train/optimizer.py:15-31: torch.optim.Adam, one group per tensor) driven through
    the reference's step form (trainer.py:116-149) — what a user who changes nothing but the three module strings gets."""
    from invr import driver
    from invr.trainer import NetworkWrapper
    bnp, _ = scene_mod.make_scene(512, 512, seed=0, cam_dist=1.8, crop=(240, 240, 32, 32))
    batch = {k: v.to(dev) for k, v in scene_mod.to_torch(bnp).items()}
    net.train()
    if hasattr(net, '_grad_arena'):
        del net._grad_arena              # (an earlier FusedAdam's persistent arena: every optimizer here starts from a plain network)
    for p_ in net.parameters():
        p_.grad = None
    wrap = NetworkWrapper(net)
    opt = driver.make_optimizer(net, fused=fused)
    opt_name = '%s.%s' % (type(opt).__module__, type(opt).__name__)
    adopted = lambda: getattr(opt, '_invr_inner', None) is not None
    for i in range(8):                   # (plan build, arena, allocator and clocks settled: the probe runs behind the eval variants)
        driver.train_step(wrap, opt, batch, i + 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        loss, _ = driver.train_step(wrap, opt, batch, i + 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    # where the iteration goes: forward / backward / optimizer step, synchronised (3 extra iterations)
    parts = [0.0, 0.0, 0.0]
    for i in range(3):
        b = dict(batch); b['iter_step'] = 50 + i
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ret, l2, _, _ = wrap(b, 0, split='train')
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        l2.mean().backward()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        parts = [parts[0] + (t2 - t1) / 3, parts[1] + (t3 - t2) / 3, parts[2] + (t4 - t3) / 3]
    net.eval()
    if hasattr(net, '_grad_arena'):
        del net._grad_arena
    for p_ in net.parameters():
        p_.grad = None
    was_adopted = adopted()
    del opt
    return {'adopted_by_fused_step': was_adopted, 'ms_per_iter': dt * 1e3, 'synchronised_ms': {'forward': parts[0] * 1e3, 'backward': parts[1] * 1e3, 'optimizer_step': parts[2] * 1e3}, 'rays_per_iter': int(batch['ray_o'].shape[1]), 'samples_per_ray': S,
            'ray_samples_per_sec': batch['ray_o'].shape[1] * S / dt, 'optimizer': opt_name,
            'parameters_updated': int(sum(p.numel() for p in net.parameters() if p.requires_grad)), 'final_loss': float(loss)}


OCC_FLOPS = 2 * (19 * 64 + 64 * 17)                    # occupancy MLP 19 -> 64 -> 17 per evaluated pair
RGB_FLOPS = {3: 2 * (70 * 64 + 64 * 64 + 64 * 3),      # colour MLP 70 -> 64 -> 64 -> 3 (body, head) per WINNING pair
             2: 2 * (70 * 64 + 64 * 3)}                # 70 -> 64 -> 3 (leg, arms)           (sum = SURVEY 8d's 22.1 k / 14.0 k)


def winner_counts(out, stats):
    """Listed pairs that won their survivor's max-occupancy merge, per part (= the pairs the colour MLP evaluated, without the
    one far-constant pair per part), from the segment counts k_winner_lists left in the workspace."""
    v = _abi.ws_views(*out['_ws'])
    g_last = (max(int(stats[0]), 1) - 1) // 4096
    return [int(x) - 1 for x in v['wcnt'][:g_last + 1].sum(0).cpu().tolist()]


def frame_batches(res, cam_dist, K, dev):
    """K frames of a synthetic sequence: the same body, K poses / body orientations / latent codes (frame 0 = the frame of the round-1..3
    bench lines).  -> (collated batches on the host, on the device)."""
    cpu, gpu = [], []
    for k in range(K):
        bnp, _ = scene_mod.make_scene(res, res, seed=0, frame=(3 + 7 * k) % 100, cam_dist=cam_dist, pose_seed=k)
        b = scene_mod.to_torch(bnp)
        cpu.append(b)
        gpu.append({kk: v.to(dev) for kk, v in b.items()})
    return cpu, gpu


def frame_set(net, batches, S, rank, world, shard_of=0, want_raw=True, capture=True, capture_exchange=True, streams=False):
    """invr.frames.FrameSet over this rank's shards of `batches`: one hipGraph replay renders all of them side by side (and, world > 1,
    exchanges their tiles with one captured all-gather).  shard_of = W renders rank 0's shard of a W-way split without any exchange."""
    fns, n_rays, keep = iframes.shard_render_fns(net, batches, S, 0 if shard_of else rank, shard_of or world, want_raw=want_raw)
    fs = iframes.FrameSet(fns, n_rays, rank=0 if shard_of else rank, world=1 if shard_of else world, device=batches[0]['ray_o'].device,
                          capture=capture, capture_exchange=capture_exchange, streams=streams)
    fs._keep = keep
    return fs


def replay_bit_exact(fs, net, S, want_raw):
    """The frames the LAST timed replay left in `fs.local` against one separate eager render of each frame (same rays, same survivor
    capacity, a scratch workspace of its own), bit for bit: rgb_map, acc_map and raw.  The headline is timed with K frames in flight;
    round 5 saw one such replay differ once (profiles/r6_replay_mismatch.md has the cause) — a differing frame must never go
    unnoticed in the line that reports the rate.  -> (bool, detail)"""
    keys = ('rgb_map', 'acc_map') + (('raw',) if want_raw else ())
    scratch = None
    bad = []
    for k, (ctx, a, ws) in enumerate(fs._keep):
        cap = fs.local[k]['_ws'][3]
        if scratch is None or scratch.numel() < ws.numel():
            scratch = torch.empty(ws.numel(), dtype=torch.uint8, device=ws.device)
        net._ws = scratch
        ref = net.render_rays(ctx, a[0], a[1], a[2], a[3], S, want_raw=want_raw, max_active=cap)
        for key in keys:
            if not torch.equal(ref[key], fs.local[k][key]):
                ne = ref[key].view(torch.int32) != fs.local[k][key].view(torch.int32)
                bad.append({'frame': k, 'tensor': key, 'differing_elements': int(ne.sum())})
    net._ws = None
    return not bad, {'frames_compared': len(fs._keep), 'tensors': list(keys), 'differing': bad}


def time_frames(fn, frames, min_time=0.0, max_regions=200):
    """ms per call of fn(): regions of exactly `frames` calls between two synchronisations, repeated until min_time seconds."""
    fn()
    torch.cuda.synchronize()
    tot, n = 0.0, 0
    while True:
        t0 = time.perf_counter()
        for _ in range(frames):
            fn()
        torch.cuda.synchronize()
        tot += time.perf_counter() - t0
        n += frames
        if tot >= min_time or n >= frames * max_regions:
            return tot / n * 1e3


def variant_lines(net, cfg, batches, dev, S, in_flight=1):
    """Driver-visible variants of the headline frames (SURVEY 8d asks for them): the strong-scaling projection from
    rank 0's shard of a W-way split on this one GPU, the yaml-default 64 samples/ray, a mid-density frame (smpl_thresh 0.1: ~3x the
    survivors), the trainable 64-byte rows instead of the row-sum tables, the dense stress frame (smpl_thresh = +inf: every ray-sample
    survives), and the wall clock of the drop-in call Renderer.render(batch) itself."""
    import copy
    from invr.renderer import Renderer
    batch = batches[0]
    n_rays = int(batch['ray_o'].shape[1])
    mean_rays = sum(int(b['ray_o'].shape[1]) for b in batches) / len(batches)
    out = {}

    def frames_ms(s, frames, shard_of=0, bs=None, min_time=0.25):
        """ms per frame of the frames `bs` (default: the headline's) rendered `in_flight` at a time by one graph replay"""
        bs = bs or batches
        fs = frame_set(net, bs, s, 0, 1, shard_of=shard_of)
        ms = time_frames(fs.replay, max(1, frames // len(bs)), min_time) / len(bs)
        assert iframes.check_overflow(fs), 'workspace overflow in a variant'
        st = [o['stats'].cpu().numpy().astype('int64') for o in fs.local]
        del fs
        torch.cuda.empty_cache()
        return ms, st

    # (1) strong-scaling projection: rank 0's tile-cyclic shard of a W-way split, one GPU, frames in flight as in the headline —
    #     and with strictly one frame at a time (the latency of a single frame)
    shard, shard1 = {}, {}
    for W in (1, 2, 4, 8):
        shard[str(W)] = frames_ms(S, 20, shard_of=W)[0]
        shard1[str(W)] = frames_ms(S, 20, shard_of=W, bs=batches[:1])[0]
    out['shard_projection'] = {
        'ms_per_frame_of_rank0_shard': shard, 'projected_speedup': {w: shard['1'] / shard[w] for w in shard},
        'frames_in_flight': len(batches),
        'one_frame_at_a_time': {'ms_per_frame_of_rank0_shard': shard1, 'projected_speedup': {w: shard1['1'] / shard1[w] for w in shard1}},
        'note': 'rank 0 of a W-way tile-cyclic split rendered on ONE GPU (full per-frame scene work included, the all-gather is not), '
                '%d frames of the sequence per hipGraph replay as in the headline; one_frame_at_a_time = frame 0 alone, one replay per '
                'frame (the latency view): the single-GPU evidence for strong scaling; the measured multi-GPU curve is the driver\'s SCALE '
                'record' % len(batches)}
    # (2) 64 samples per ray (configs/inb/inb_377.yaml default)
    ms, _ = frames_ms(64, 20)
    out['samples_64'] = {'ms_per_frame': ms, 'ray_samples_per_sec': mean_rays * 64 / (ms * 1e-3), 'samples_per_ray': 64}
    # (3) a mid-density frame: smpl_thresh 0.1 (inb_lan.yaml's value) keeps ~3x the survivors of inb_377's 0.05
    mcfg = copy.deepcopy(cfg)
    mcfg['smpl_thresh'] = 0.1
    net.cfg = mcfg
    try:
        ms, st = frames_ms(S, 12)
    finally:
        net.cfg = cfg
    out['mid_density'] = {'ms_per_frame': ms, 'ray_samples_per_sec': mean_rays * S / (ms * 1e-3), 'smpl_thresh': 0.1,
                          'active_fraction': float(sum(int(s[0]) for s in st)) / (mean_rays * S * len(st)),
                          'pairs_per_active_sample': float(sum(int(s[1:6].sum()) for s in st)) / max(1, sum(int(s[0]) for s in st))}
    # (3a) the survey's own camera (3 m): the body fills a third of the frame, fewer rays hit its box, a larger share of
    #      their samples survives the cull — the headline's 1.8 m is the camera at which the body fills the 512 x 512 frame
    _, bs3 = frame_batches(512, 3.0, len(batches), dev)
    ms, st = frames_ms(S, 20, bs=bs3)
    rays3 = sum(int(b['ray_o'].shape[1]) for b in bs3) / len(bs3)
    na3 = float(sum(int(s_[0]) for s_ in st)) / len(st)
    out['survey_cam'] = {'cam_dist_m': 3.0, 'ms_per_frame': ms, 'rays': int(round(rays3)), 'ray_samples_per_sec': rays3 * S / (ms * 1e-3),
                         'active_samples': int(na3), 'active_fraction': na3 / (rays3 * S), 'survivors_per_sec': na3 / (ms * 1e-3),
                         'pairs_per_active_sample': float(sum(int(s_[1:6].sum()) for s_ in st)) / max(1.0, na3 * len(st)),
                         'note': 'SURVEY.md 8(d)\'s camera: 3 m from the body (f = 555 px at 512^2): only the rays that hit the body AABB are rendered, as in '
                                 'the reference (if_nerf_data_utils.py:298-308); same sequence of %d poses, %d frames per graph replay' % (len(bs3), len(bs3))}
    del bs3
    # (3b) cfg.aggr = 'mean' (inb_part_network_multiassign.py:236-239): every listed pair needs its colour, not only each survivor's winner
    acfg = copy.deepcopy(cfg)
    acfg['aggr'] = 'mean'
    net.cfg = acfg
    try:
        ms, _ = frames_ms(S, 12)
    finally:
        net.cfg = cfg
    out['aggr_mean'] = {'ms_per_frame': ms, 'ray_samples_per_sec': mean_rays * S / (ms * 1e-3),
                        'note': 'the mean merge: k_part_rgb_all over all listed pairs (about twice the winners) + k_winner_lists<MEAN>'}
    # (3c) cfg.aggr = 'dist' (:240-244, round 5): part_dist of EVERY (survivor, part) from a brute-force pass (k_knn_pdist) on top of the mean flow
    acfg = copy.deepcopy(cfg)
    acfg['aggr'] = 'dist'
    net.cfg = acfg
    try:
        ms, _ = frames_ms(S, 12)
    finally:
        net.cfg = cfg
    out['aggr_dist'] = {'ms_per_frame': ms, 'ray_samples_per_sec': mean_rays * S / (ms * 1e-3),
                        'note': 'the distance-weighted merge: aggr_mean + an exact brute-force 4-NN of all five parts per survivor (non-default mode, not tuned)'}
    # (4) the 64-byte trainable rows (what a training-mode forward reads) instead of the derived row-sum tables
    old = cfg.get('eval_row_sums', True)
    cfg['eval_row_sums'] = False
    try:
        ms, _ = frames_ms(S, 12)
    finally:
        cfg['eval_row_sums'] = old
    out['full_rows'] = {'ms_per_frame': ms, 'ray_samples_per_sec': mean_rays * S / (ms * 1e-3),
                        'table_bytes_per_pair': PAIR_TABLE_BYTES, 'note': 'k_part_encode: 16 levels x 8 corners x 64-byte rows per pair'}
    # (5) dense stress: every ray-sample survives the cull (113 M evaluated pairs, 28 GB workspace), 3 eager frames of frame 0
    ro, rd, nr, fr = (batch[k][0].contiguous() for k in ('ray_o', 'ray_d', 'near', 'far'))
    dcfg = copy.deepcopy(cfg)
    dcfg['smpl_thresh'] = 1e9
    net.cfg = dcfg
    try:
        dctx = net.prepare(batch)
        net._ws = None
        ms = time_frames(lambda: net.render_rays(dctx, ro, rd, nr, fr, S, want_raw=True), 3, 0.0)
        st = net.render_rays(dctx, ro, rd, nr, fr, S, want_raw=True)['stats'].cpu().numpy().astype('int64')
    finally:
        net.cfg = cfg
        net._ws = None                                       # give the 28 GB back
    out['dense_stress'] = {'ms_per_frame': ms, 'ray_samples_per_sec': n_rays * S / (ms * 1e-3), 'frames': 3,
                           'active_samples': int(st[0]), 'evaluated_pairs': int(st[1:6].sum())}
    # (6) the drop-in call: Renderer.render(batch) as run.py / the evaluator call it (wall clock, host side included)
    api = {}
    for to_cpu, pin, key in ((False, False, 'eval_to_cpu_false_ms'), (True, False, 'eval_to_cpu_true_ms'), (True, True, 'eval_to_cpu_true_pinned_ms')):
        r = Renderer(net)
        r.eval_to_cpu, r.pin_host = to_cpu, pin
        b = dict(batch)
        r.render(b)
        r.render(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ret = r.render(b)
            _ = ret['rgb_map'], ret['acc_map']              # what the reference's evaluator / visualizers read (evaluators/if_nerf.py:77)
        torch.cuda.synchronize()
        api[key] = (time.perf_counter() - t0) / 3 * 1e3
    # ... and with frames in flight ACROSS render() calls (Renderer.in_flight lanes, round 5): the caller submits the frames of the
    # sequence one render(batch) at a time — every batch a new dict with its own volume dimensions, nothing captured — and reads
    # frame f's maps after submitting frames f+1 .. f+7, as driver.run_evaluate does
    from collections import deque
    del r, ret
    net._ws = None
    torch.cuda.synchronize()
    torch.cuda.empty_cache()             # (the lanes allocate from their own streams' pools: give them the blocks the variants above cached)
    api['reserved_gb_before_in_flight'] = torch.cuda.memory_reserved() / 1e9
    r = Renderer(net)                      # ONE renderer, as a host has: its lanes keep their workspaces / raw buffers between the two modes
    r.in_flight = 8
    for to_cpu, key in ((False, 'in_flight8_eval_to_cpu_false_ms'), (True, 'in_flight8_eval_to_cpu_true_ms')):
        r.eval_to_cpu = to_cpu
        q = deque()

        def sweep(n):
            for i in range(n):
                q.append(r.render(dict(batches[i % len(batches)])))
                while len(q) >= r.in_flight:
                    o = q.popleft()
                    _ = o['rgb_map'], o['acc_map']
            while q:
                o = q.popleft()
                _ = o['rgb_map'], o['acc_map']
            torch.cuda.synchronize()
        # warm until a whole sweep allocates nothing: lanes take their workspaces / raw buffers as the frames come (grow-only survivor
        # bound, a second raw buffer while a caller still holds the lane's previous dict), and behind the variants above — 28 GB
        # blocks given back to the driver — a multi-GB hipMalloc takes 0.1 - 0.5 s on this runtime: round 6's first line carried
        # one of those in the timed sweep (13 ms per frame; tools/exp_api_order.py shows the steady state is reached in 2 - 3 sweeps)
        for _warm in range(8):
            before = torch.cuda.memory_reserved()
            sweep(2 * len(batches))
            if _warm >= 1 and torch.cuda.memory_reserved() == before:
                break
        api[key.replace('_ms', '_warm_sweeps')] = _warm + 1
        t0 = time.perf_counter()
        sweep(4 * len(batches))
        api[key] = (time.perf_counter() - t0) / (4 * len(batches)) * 1e3
        api[key.replace('_ms', '_reserved_gb')] = torch.cuda.memory_reserved() / 1e9
        api[key.replace('_ms', '_lane_gb')] = [round(((l.ws.numel() if l.ws is not None else 0) + (l.raw_buf.numel() * 4 if l.raw_buf is not None else 0)) / 1e9, 2)
                                               for l in r._lanes]
    r.flush(release=True)
    del r
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    api['outputs'] = ['acc_map', 'occ', 'raw', 'rgb_map']
    api['note'] = ('Renderer.render(batch): eager launches + statistics read-back, one frame at a time (the maps are read after every call); '
                   'in_flight8_* = the same call with Renderer.in_flight = 8 over the frames of the sequence, maps read 7 calls later '
                   '(driver.run_evaluate\'s loop); eval_to_cpu=True is the reference contract '
                   '(inb_renderer.py:199-200 moves every output to the host: raw + occ = %.0f MB), into ordinary host tensors as torch\'s .cpu() '
                   'gives (`_pinned` = page-locked ones: a faster copy, but uncached for the CPU on this platform — 4.6 ms per pass over a 3 MB map)' % (n_rays * S * 20 / 1e6))
    out['api_frame'] = api
    torch.cuda.empty_cache()
    return out


def main_train(args):
    """--train: the training iteration of the path as the bench step (BASELINE configs[4]: 1024 rays x 128 samples per rank,
    inb_377 defaults, forward + backward fused HIP + dense Adam over all 286 M parameters; --train-config lan = configs[3]:
    inb_lan.yaml (smpl_thresh 0.1, lr 1e-3, pair_loss_weight 1e-4), 64x64 patches x 64 samples, the reference's epoch /
    iteration schedule, optional wall-clock --budget).  N ranks = data parallel (invr.dist_train): every rank trains on its own
    patches (per-rank work fixed -> "scaling": "weak"), gradients are averaged with one all-reduce per gradient block,
    overlapped with the backward."""
    from invr import driver, dist_train
    from invr.trainer import NetworkWrapper
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world)
    dev_index = int(os.environ.get('INVR_FORCE_DEVICE', local_rank))
    backend = os.environ.get('INVR_DIST_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
    lan = args.train_config == 'lan'
    S = 64 if lan else args.samples
    side = 64 if lan else 32
    kw = dict(N_samples=S)
    if args.table_log2:
        kw['table_log2'] = args.table_log2
    if lan:
        kw.update(smpl_thresh=0.1, pair_loss_weight=1e-4)
    cfg = make_cfg(**kw)
    net = build_model(cfg, dev).train()
    if world > 1:
        dist_train.broadcast_parameters(net)
    n_params = sum(p.numel() for p in net.parameters() if p.requires_grad)
    wrap = NetworkWrapper(net)
    opt = driver.make_optimizer(net, lr=1e-3 if lan else 5e-4, eps=1e-15)
    sched = driver.ExponentialLR(opt, decay_epochs=1000, gamma=0.1)
    red = dist_train.attach(opt) if world > 1 else None
    # a small pool of patches per rank, resident in HBM (the reference's loader prefetches 8 batches deep, trainer.py:83-88)
    pool = []
    for k in range(8):
        # patch corners spread over the figure (head .. legs, torso .. arms), as random crops of the reference's sampler do
        c = (90 + 70 * ((k + 3 * rank) % 5), 190 + 32 * ((2 * k + rank) % 4))
        bnp, _ = scene_mod.make_scene(512, 512, seed=0, frame=(7 * k + 13 * rank) % 100, cam_dist=1.8, crop=(c[0], c[1], side, side))
        pool.append({kk: v.to(dev) for kk, v in scene_mod.to_torch(bnp).items()})
    rays = sum(int(b['ray_o'].shape[1]) for b in pool) / len(pool)         # mean rays per iteration (patches near the AABB edge lose a few)
    it = [0]

    def step():
        b = dict(pool[it[0] % len(pool)])
        it[0] += 1
        return driver.train_step(wrap, opt, b, 2 + (it[0] % 400))[0]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    for _ in range(max(args.warmup, 2)):
        loss = step()
    fence()
    region = []
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        fence()
        region.append(time.perf_counter() - t0)
        total = time.perf_counter() - t_start
        enough = (total >= args.budget) if args.budget else (sum(region) >= args.min_time or len(region) >= 1000)
        if world > 1:
            flag = torch.tensor([1 if enough else 0], device=dev)
            dist.broadcast(flag, 0)
            enough = bool(int(flag.item()))
        if enough:
            break
    dt = sum(region) / len(region)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the dominant kernel of the iteration: the dense Adam step (HBM-bound), timed with events on the launch stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    adam_ms = []
    for _ in range(5):
        b = dict(pool[0]); b['iter_step'] = 5
        ret, l2, _, _ = wrap(b, 0, split='train')
        opt.zero_grad(set_to_none=True)
        l2.mean().backward()
        if red is not None:
            red.wait()
        e0.record(); opt.step(); e1.record()
        torch.cuda.synchronize()
        adam_ms.append(e0.elapsed_time(e1))
    adam_ms = sorted(adam_ms)[len(adam_ms) // 2]
    stats = wrap.renderer.last_stats.cpu().numpy().astype('int64')
    if rank == 0:
        ms = dt / args.steps * 1e3
        total_rs = rays * S * world
        # bytes the timed Adam launch moved: p, m, v read+write (24 B) + gradient (dense 4 B, row-scalar tables 0.25 B) for every
        # trainable tensor (all of them take every step, as in the reference: zero gradients, not None, for an absent part)
        table_ids = {id(t) for t in opt.arena.tables}
        adam_bytes = 0
        for p_ in net.parameters():
            if not p_.requires_grad:
                continue
            adam_bytes += p_.numel() * 24 + (p_.numel() // 4 if id(p_) in table_ids else p_.numel() * 4)
        line = {
            'metric': 'ray-samples/sec (training iteration: forward + backward + Adam)', 'value': total_rs * args.steps / dt, 'unit': 'ray-samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms, 'repeats': len(region), 'timed_region_s': sum(region),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': ('configs[3]: inb_lan.yaml training (smpl_thresh 0.1, lr 1e-3), 64x64 patch x 64 samples per iteration' if lan else
                             'configs[4]: inb_377 training, %d rays x %d samples per rank per iteration' % (round(rays), S)) +
                            ', full-size model, forward + backward fused HIP + dense Adam',
                'rays_per_rank': float(rays), 'samples_per_ray': S, 'ray_samples_per_step': int(total_rs), 'parameters_updated': int(n_params),
                'active_samples_rank0': int(stats[0]), 'pairs_per_part_rank0': [int(v) for v in stats[1:6]],
                'optimizer': type(opt).__name__, 'iterations_timed': args.steps * len(region),
                'parallelism': 'dp%d: full replicas, row-scalar table gradients (%.0f MB) + %.1f MB small tensors averaged per iteration, '
                               'all-reduce overlapped with the backward' % (world, 4e-6 * sum(e.row_grad().numel() for e in opt.arena.embedders),
                                                                           4e-6 * opt.arena.flat.numel()),
                'final_loss': float(loss),
            },
            'roofline': {'kernel': 'k_adam (dense Adam over every parameter, 1 launch/step)', 'bound': 'hbm',
                         'achieved': adam_bytes / (adam_ms * 1e-3) / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                         'frac': adam_bytes / (adam_ms * 1e-3) / HBM_PEAK, 'traffic': None,
                         'algorithmic_bytes_per_launch': int(adam_bytes), 'kernel_ms_per_launch': adam_ms,
                         'note': '24 B per parameter (param, exp_avg, exp_avg_sq read + write) + gradient: 4 B dense, 0.25 B for the '
                                 'row-scalar table gradients; torch events on the launch stream around the step'},
            'cpu_baseline': None,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train', action='store_true', help='bench the training iteration instead of the eval frame (see main_train)')
    ap.add_argument('--train-config', choices=['377', 'lan'], default='377', help='--train: configs[4] (377) or configs[3] (lan)')
    ap.add_argument('--budget', type=float, default=0.0, help='--train: keep training for this many seconds of wall clock (configs[3]: 300)')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--res', type=int, default=512)
    ap.add_argument('--samples', type=int, default=128)
    ap.add_argument('--table-log2', type=int, default=None, help='debug: cap log2_hashmap_size')
    ap.add_argument('--dense', action='store_true', help='stress variant: smpl_thresh=+inf (every sample active)')
    ap.add_argument('--no-raw', action='store_true', help='do not materialise raw/occ (N x 20 B)')
    ap.add_argument('--cam-dist', type=float, default=1.8, help='camera distance (m); 1.8 -> 97.6%% of the 512x512 pixels hit the body AABB')
    ap.add_argument('--cpu-rays', type=int, default=4096, help='rays of the CPU-baseline chunk (BASELINE.md §3: one 4096-ray chunk)')
    ap.add_argument('--min-time', type=float, default=1.0, help='repeat the timed K-step region until this many seconds are timed in total')
    ap.add_argument('--full-rows', action='store_true', help='read the trainable 64-byte table rows instead of the eval-mode row-sum tables')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--train-iters', type=int, default=30, help='iterations of the informational training-step probe (0 = skip; N=1 only)')
    ap.add_argument('--no-graph', action='store_true', help='launch the ~25 kernels of a frame eagerly instead of replaying one captured hipGraph')
    ap.add_argument('--in-flight', type=int, default=10, help='frames of the sequence rendered side by side by one hipGraph replay (invr.frames); reduced to a divisor of --steps; 1 = strictly one frame at a time')
    ap.add_argument('--no-variants', action='store_true', help='skip the S=64 / dense / full-row / shard-projection / API-frame variants of the default line')
    ap.add_argument('--shard-of', type=int, default=0, help='debug (1 GPU): render only rank 0\'s ray shard of a W-way split')
    args = ap.parse_args()
    if args.train:
        return main_train(args)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    # debugging aids for a 1-GPU box: INVR_FORCE_DEVICE pins every rank to one GPU and
    # INVR_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    dev_index = int(os.environ.get('INVR_FORCE_DEVICE', local_rank))
    backend = os.environ.get('INVR_DIST_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    kw = dict(N_samples=args.samples)
    if args.table_log2:
        kw['table_log2'] = args.table_log2
    if args.dense:
        kw['smpl_thresh'] = 1e9
    cfg = make_cfg(**kw)
    cfg['eval_row_sums'] = not args.full_rows
    net = build_model(cfg, dev)
    n_params = sum(p.numel() for p in net.parameters())
    S = args.samples
    want_raw = not args.no_raw
    # Frames in flight: K frames of the sequence (K poses) are rendered by ONE hipGraph replay, as K parallel branches (invr.frames);
    # a step is still one frame, and a timed region is exactly --steps frames = --steps / K replays, so K divides --steps.
    K = max(1, min(args.in_flight, args.steps))
    while args.steps % K:
        K -= 1
    if args.dense or args.no_graph:
        K = 1
    batches_cpu, batches = frame_batches(args.res, args.cam_dist, K, dev)
    batch_cpu, batch = batches_cpu[0], batches[0]
    n_rays = batch['ray_o'].shape[1]
    rays_per_frame = [int(b['ray_o'].shape[1]) for b in batches]
    mean_rays = sum(rays_per_frame) / K

    # frame 0's shard for the eager per-stage profile (below) — the timed frames are rendered through the FrameSet
    idx = idist.tile_indices(n_rays, rank, args.shard_of or world, device=dev)
    ro, rd = batch['ray_o'][0][idx].contiguous(), batch['ray_d'][0][idx].contiguous()
    nr, fr = batch['near'][0][idx].contiguous(), batch['far'][0][idx].contiguous()
    ctx = net.prepare(batch)

    def render():
        return net.render_rays(ctx, ro, rd, nr, fr, S, want_raw=want_raw)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # N > 1: every rank renders its tile-cyclic shard of the K frames; ONE all-gather of the K frames' [r, g, b, acc] tiles and the K
    # index_selects are part of the same captured graph: one replay = K complete frames on every rank, no per-frame host work.
    use_graph = not args.no_graph
    fs = None
    # Modes, best first; every rank must end up in the same one (the exchange is a collective), so each attempt's outcome is agreed on
    # with an all-reduce: (1) renders + exchange captured — the captured exchange must also reproduce this rank's own rows on every
    # rank (a runtime that mis-replays captured collectives fails here); (2) renders captured, the exchange issued from the host
    # once per K frames; (3) eager launches (the same work, launch-bound).
    modes = ([('graph+exchange', True, True), ('graph', True, False)] if use_graph else []) + [('eager', False, False)]
    if world == 1:
        modes = [m for m in modes if m[0] != 'graph']
    for name, cap, cap_x in modes:
        ok = 1
        try:
            fs = frame_set(net, batches, S, rank, world, shard_of=args.shard_of, want_raw=want_raw, capture=cap, capture_exchange=cap_x)
            if world > 1 and cap_x:
                fs.replay()
                ok = 1 if fs.own_rows_match() else 0
                if not ok:
                    sys.stderr.write('captured all-gather did not reproduce the local rows\n')
        except Exception as e:                       # keep the bench alive: the next mode measures the same work
            sys.stderr.write('%s failed (%s)\n' % (name, e))
            ok = 0
        if world > 1:
            flag = torch.tensor([ok], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            mode_name = name if world > 1 else ('graph' if cap else 'eager')          # (one GPU: nothing to exchange)
            break
        fs = None
        torch.cuda.empty_cache()
        if name == 'eager':
            raise RuntimeError('no render mode worked')
        sys.stderr.write('falling back from %s\n' % name)
    use_graph = fs.graph is not None
    for _ in range(max(1, -(-args.warmup // K))):
        fs.replay()
    fence()
    # The timed region is EXACTLY K steps (frames) between two fences.  A frame takes ~2 ms, so one region is a few tens of
    # milliseconds: the region is repeated (each repeat again exactly K steps between fences) until >= --min-time seconds
    # have been timed, and the reported time per step is the mean over all repeats.
    region = []
    while True:
        t0 = time.perf_counter()
        for _ in range(args.steps // K):
            fs.replay()
        fence()
        region.append(time.perf_counter() - t0)
        enough = sum(region) >= args.min_time or len(region) >= 1000
        if world > 1:
            flag = torch.tensor([1 if enough else 0], device=dev)
            dist.broadcast(flag, 0)
            enough = bool(int(flag.item()))
        if enough:
            break
    repeats = len(region)
    dt = sum(region) / repeats
    assert iframes.check_overflow(fs), 'workspace overflow'
    for k in range(K):
        assert fs.full[k] is not None and fs.full[k].shape[0] == (rays_per_frame[k] if not args.shard_of else fs.full[k].shape[0])
        assert bool(torch.isfinite(fs.full[k]).all())
    frame_stats = [o['stats'].cpu().numpy().astype('int64') for o in fs.local]
    # what the last timed replay produced, against separate renders of the same frames, bit for bit (every rank its own shards)
    bit_exact, bit_detail = replay_bit_exact(fs, net, S, want_raw)
    if world > 1:
        flag = torch.tensor([1 if bit_exact else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        bit_exact = bool(int(flag.item()))
    # the exchange alone (N > 1): the captured all-gather + index_selects replayed without the renders
    exchange_ms = None
    exchange_captured = bool(fs.exchange and fs.exchange_captured)
    if world > 1 and fs.exchange and fs.exchange_captured:
        try:
            gx = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gx, capture_error_mode='thread_local'):
                fs._exchange()
            fence()
            t0 = time.perf_counter()
            for _ in range(50):
                gx.replay()
            fence()
            exchange_ms = (time.perf_counter() - t0) / 50 * 1e3
        except Exception as e:
            sys.stderr.write('exchange-only timing failed (%s)\n' % e)
    # per-stage HIP-event times: a few extra eager frames (frame 0) outside the timed region (event records are not
    # replayable graph nodes)
    net._ws = None
    _abi.profile_enable(True)
    _abi.profile_read()
    for _ in range(3):
        out = render()
    torch.cuda.synchronize()
    _abi.profile_enable(False)
    stage_ms, n_prof = _abi.profile_read()
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stats = out['stats'].cpu().numpy().astype('int64')              # frame 0, this rank's shard (the eager profile frames)
    assert stats[6] == 0, 'workspace overflow'
    stats_sum = sum(frame_stats)                                     # the K timed frames, this rank's shards
    if world > 1:
        st = torch.from_numpy(stats_sum).to(dev)
        dist.all_reduce(st)
        stats_sum = st.cpu().numpy()
    stats_all = stats_sum / K                                        # per frame, all ranks

    if rank == 0:
        total_samples = mean_rays * S                    # per step (frame): the K frames of the sequence differ by a few rays
        ms_per_step = dt / args.steps * 1e3
        value = total_samples * args.steps / dt          # dt = mean duration of one K-step region (= steps / K replays of K frames)
        pairs_local = int(stats[1:6].sum())
        winners = winner_counts(out, stats)              # rank 0's shard
        n_rgb = [len(pn.rgb.linears) for pn in net.tpose_human.part_networks]
        per = max(n_prof, 1)
        enc_ms = sum(stage_ms['encode_%d' % p] for p in range(5)) / per
        mlp_ms = sum(stage_ms['mlp_%d' % p] for p in range(5)) / per
        knn_ms = stage_ms['knn'] / per
        enc_bytes = pairs_local * (PAIR_TABLE_BYTES if args.full_rows else PAIR_TABLE_BYTES // 16)   # 16 levels x 8 corners x 64 B | 4 B
        # part MLPs: occupancy MLP for every listed pair + colour MLP for the winning pair of every survivor (+ the far constants)
        mlp_flops = OCC_FLOPS * pairs_local + sum(RGB_FLOPS[n_rgb[p]] * (winners[p] + 1) for p in range(5))
        mlp_flops_ref = sum(int(stats[1 + p]) * (OCC_FLOPS + RGB_FLOPS[n_rgb[p]]) for p in range(5))      # what evaluating every pair costs
        knn_flops = int(stats[0]) * 62000                                                                # brute-force 4-NN of the reference, SURVEY 8(d)
        # per-kernel counters of the same command from the PMC passes (tools/prof_all.sh -> tools/prof_summary.py -> profiles/):
        # HBM / Infinity-Cache bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction), busy and wait fractions
        traffic, counters = {}, {}
        headline = world == 1 and not args.dense and args.table_log2 is None and args.res == 512 and S == 128 and not args.full_rows \
            and not args.shard_of
        tf, cf = os.path.join(ROOT, 'profiles', 'hbm_traffic_per_launch.json'), os.path.join(ROOT, 'profiles', 'kernel_counters.json')
        if headline and os.path.exists(tf):
            traffic = json.load(open(tf))
        if headline and os.path.exists(cf):
            counters = json.load(open(cf))
        meta = counters.pop('_meta', {}) if isinstance(counters, dict) else {}
        traffic.pop('_meta', None)
        # the profiler reports template instantiations by their full name: the eval frame's are <false> (arg-max merge)
        for d in (traffic, counters):
            for k in ('k_winner_lists', 'k_part_rgb_all'):
                for suffix in ('<false>', '<0>'):          # (k_winner_lists is <0> since round 5's merge modes)
                    if k not in d and k + suffix in d:
                        d[k] = d.pop(k + suffix)
        counters_stale = bool(traffic or counters) and meta.get('csrc_digest') != csrc_digest()
        src = ('profiles/ (rocprofv3 --pmc passes of this command, tools/prof_all.sh; counters cannot be collected inside the timed run)'
               + (' — STALE: measured on other kernel sources (digest %s, now %s)' % (meta.get('csrc_digest'), csrc_digest()) if counters_stale else '')
               if traffic else None)
        mlp_kernels = ('k_part_occ_all', 'k_winner_lists', 'k_part_rgb_all')
        mlp_traffic = sum(traffic.get(k, 0) for k in mlp_kernels) if all(k in traffic for k in mlp_kernels) else None
        # the byte model of SURVEY 8(d) / BASELINE.md §4 for the whole frame (a MODEL, not a bound: see path_roofline.note)
        tab_b = PAIR_TABLE_BYTES if args.full_rows else PAIR_TABLE_BYTES // 16
        path_bytes = (tab_b + 512 + 64) * pairs_local + 1920 * int(stats[0]) + (32 + (20 if want_raw else 0)) * int(ro.shape[0]) * S + 48 * int(ro.shape[0])
        path_traffic = (sum(v for k, v in traffic.items() if k != 'k_row_sums') if traffic else None)      # (row sums: once per weight version)
        tfl = lambda fl, ms: fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        binding = lambda k: counters.get(k)
        line = {
            'metric': 'ray-samples/sec', 'value': value, 'unit': 'ray-samples/s', 'n_gpus': world,
            'survivors_per_sec': float(stats_all[0]) * args.steps / dt,      # co-headline: the samples that survive the cull and reach the networks
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step, 'repeats': repeats, 'timed_region_s': sum(region), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'configs[1]: ZJU-MoCap-377-shaped synthetic frame, inb_377 defaults (full 1.09 GB tables), '
                            '%dx%d, %d samples/ray%s' % (args.res, args.res, S, ', DENSE stress (smpl_thresh=inf)' if args.dense else ''),
                'rays': int(round(mean_rays)), 'rays_per_frame': rays_per_frame, 'samples_per_ray': S, 'ray_samples_per_step': int(total_samples),
                'active_samples': int(stats_all[0]), 'active_samples_per_frame_rank0': [int(s[0]) for s in frame_stats],
                'active_fraction': float(stats_all[0]) / total_samples,
                'survivors_per_sec': float(stats_all[0]) * args.steps / dt,
                'pairs_per_part': [int(v) for v in stats_all[1:6]],
                'pairs_per_active_sample': float(stats_all[1:6].sum()) / max(int(stats_all[0]), 1),
                'colour_mlp_pairs_per_part_rank0': winners,
                'parameters': int(n_params), 'raw_occ_materialised': want_raw, 'occ_is_raw_channel_3_view': True, 'hip_graph': use_graph, 'frames_in_flight': K,
                'frames': '%d frames of a synthetic sequence (same body, %d poses / orientations / latent codes; frame 0 = the frame of the '
                          'round-1..3 lines) rendered side by side by ONE hipGraph replay (parallel branches, invr.frames.FrameSet); a step is '
                          'one frame, every frame does all of its per-frame scene work' % (K, K),
                'render_mode': mode_name,
                'parallelism': ('tile-cyclic ray shards x%d, full replicas, 1 all-gather per %d frames ' % (world, K)) +
                               ('inside the same graph replay' if exchange_captured else 'issued from the host behind the %s' % ('graph replay' if use_graph else 'eager launches'))
                               if world > 1 else 'one GPU: whole frames',
                'exchange_only_ms_per_replay': exchange_ms, 'exchange_captured_in_graph': exchange_captured,
                'rays_per_sec': mean_rays * args.steps / dt,
                'note': 'value counts every ray-sample of the frames; %.1f %% of them survive the near-surface cull.  The camera sits at %.1f m '
                        '(default) because BASELINE configs[1] is a 512 x 512 frame of rays: at SURVEY 8d\'s 3 m only ~1/3 of the pixels hit the '
                        'body AABB (the reference renders only those), so the workload would be a third of a frame — survey_cam reports that '
                        'camera, survivors_per_sec the rate of the samples that reach the networks, mid_density the same sequence at '
                        'smpl_thresh 0.1' % (100.0 * float(stats_all[0]) / total_samples, args.cam_dist),
            },
            # dominant roofline-bound stage: the tiny MLPs of all five parts on the fp32 matrix cores
            'roofline': {
                'kernel': 'part MLPs = k_part_occ_all (19-64-17 for every listed pair) + k_winner_lists + k_part_rgb_all (70-64(-64)-3 for the '
                          'pair that wins each survivor\'s max-occupancy merge); occupancy MLP on v_mfma_f32_16x16x4_f32, the colour MLP\'s two '
                          '64-wide layers on v_mfma_f32_16x16x32_bf16 as 3-way bf16 splits (6 products, fp32 accumulation: fp32 accuracy)', 'bound': 'mfma',
                'achieved': tfl(mlp_flops, mlp_ms), 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': tfl(mlp_flops, mlp_ms) / 157.3,
                'traffic': mlp_traffic, 'traffic_source': src,
                'algorithmic_flops_per_launch': int(mlp_flops), 'kernel_ms_per_launch': mlp_ms,
                'flops_if_every_pair_ran_both_mlps': int(mlp_flops_ref),
                'counters': {k: binding(k) for k in mlp_kernels} if counters else None,
                'note': 'algorithmic = 4.6 kFLOP x listed pairs + 17.5 k (body, head) / 9.3 kFLOP (leg, arms) x winning pairs on rank 0 '
                        '(SURVEY 8d\'s 22.1 k / 14.0 k split into its two MLPs; the reference evaluates both for every pair and discards the '
                        'colour of all but the arg-max part, inb_part_network_multiassign.py:253-256); peak = the fp32-in MFMA peak (= fp32 vector peak '
                        'on gfx950), kept as the yardstick for the ALGORITHMIC fp32 FLOPs although the colour layers now execute each algorithmic '
                        'product as 6 bf16 products at 16x the fp32 MFMA rate (an fp32-accurate ceiling of 2.5 PFLOP/s / 6 = 417 TFLOP/s for those '
                        'layers); time = HIP events on the launch stream around the three launches of the stage',
            },
            'roofline_other': [
                {'kernel': 'k_knn_pairs (largest single kernel; exact per-part 4-NN)', 'bound': 'valu-issue (no HBM / MFMA roofline applies)',
                 'achieved': tfl(knn_flops, knn_ms), 'peak': 157.3, 'unit': 'TFLOP/s (brute-force-equivalent)',
                 'frac': None, 'traffic': traffic.get('k_knn_pairs'), 'kernel_ms_per_launch': knn_ms, 'counters': binding('k_knn_pairs'),
                 'note': 'achieved = brute-force-EQUIVALENT rate: the search the reference runs costs 6890 vertices x ~9 FLOP = 62 kFLOP per '
                         'survivor (SURVEY 8d); the cluster-pruned exact search executes roughly a tenth of it, so the figure can exceed the '
                         'vector peak and no fraction is quoted; the binding resource is VALU issue (counters.valu_busy)'},
                {'kernel': 'k_part_encode_rs_xcd (hash-grid gathers through the eval-mode row-sum tables)' if not args.full_rows
                           else 'k_part_encode (64-byte table rows)',
                 'bound': 'valu-issue + L2-miss latency (tables are L2 / Infinity-Cache resident)' if not args.full_rows else 'hbm',
                 'measured_bytes_per_launch': traffic.get('k_part_encode_rs_xcd') if not args.full_rows else None,
                 'measured_GBps': (traffic['k_part_encode_rs_xcd'] / (enc_ms * 1e-3) / 1e9) if (not args.full_rows and enc_ms > 0 and 'k_part_encode_rs_xcd' in traffic) else None,
                 'kernel_ms_per_launch': enc_ms, 'counters': binding('k_part_encode_rs_xcd') if not args.full_rows else None,
                 'nominal_model': {'bytes_per_launch': int(enc_bytes), 'GBps': enc_bytes / (enc_ms * 1e-3) / 1e9 if enc_ms > 0 else 0.0,
                                   'frac_of_hbm_peak': (enc_bytes / (enc_ms * 1e-3) / HBM_PEAK) if enc_ms > 0 else 0.0,
                                   'note': 'SECONDARY figure: SURVEY 8d\'s per-pair bytes (16 levels x 8 corners x %d B) x pairs — gathers that hit '
                                           'L1 / L2 are counted, so this is not HBM traffic and can exceed 1 (64-byte rows: 1.96 in round 1)'
                                           % (64 if args.full_rows else 4)}},
            ],
            'path_roofline': {
                'bound': 'issue (VALU / MFMA) — no kernel of the frame is HBM-bound',
                'measured_bytes_per_step': path_traffic, 'measured_GBps': (path_traffic / (ms_per_step * 1e-3) / 1e9) if path_traffic else None,
                'measured_frac_of_hbm_peak': (path_traffic / (ms_per_step * 1e-3) / HBM_PEAK) if path_traffic else None,
                'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'traffic_source': src,
                'nominal_model': {
                    'bytes_per_step': int(path_bytes), 'GBps': path_bytes / (ms_per_step * 1e-3) / 1e9, 'frac_of_hbm_peak': path_bytes / (ms_per_step * 1e-3) / HBM_PEAK,
                    'formula': '(%d + 512 + 64) B x evaluated pairs + 1920 B x survivors + (32%s) B x ray-samples + 48 B x rays (SURVEY 8d, rank 0 shard)'
                               % (tab_b, ' + 20' if want_raw else ''),
                    'note': 'SECONDARY figure, not a bound: the byte model counts every gather as memory traffic; the measured HBM / Infinity-Cache '
                            'bytes are about a third of it (rows are reused from L1 / L2), and with the 64-byte rows the model prices the encoder '
                            'above the HBM peak'},
                'note': 'the frame is issue-bound: KNN = VALU, part MLPs = MFMA + transcendental issue, encoder = VALU index math + L2-miss '
                        'latency; see roofline / roofline_other counters'},
            'replay_bit_exact': bit_exact, 'replay_check': bit_detail,
            'counters_stale': counters_stale, 'csrc_digest': csrc_digest(),
            'stage_ms_per_step': {k: v / per for k, v in stage_ms.items()},
            'stage_note': 'HIP-event stage times of frame 0 rendered ALONE (eager launches after the timed region); with %d frames in flight '
                          'the stages of different frames overlap, so their sum exceeds ms_per_step' % K,
        }
        if world == 1 and not args.no_variants and headline:
            try:
                del fs
                out = None                   # (the eager profile frame holds a full-capacity workspace: 37.6 GB)
                net._ws = None
                torch.cuda.empty_cache()
                line.update(variant_lines(net, cfg, batches, dev, S, K))
            except Exception as e:          # informational: never lose the bench line over a variant
                line['variants_error'] = repr(e)
        if world == 1 and args.train_iters > 0 and not args.shard_of:
            try:
                line['train_step'] = train_probe(net, dev, S, args.train_iters)
            except Exception as e:          # informational only: never lose the bench line over it
                line['train_step'] = {'error': repr(e)}
            try:
                torch.cuda.empty_cache()
                line['api_train_step'] = train_probe(net, dev, S, args.train_iters, fused=False)
                line['api_train_step']['note'] = ('NetworkWrapper + the reference\'s own optimizer construction (torch.optim.Adam, 186 one-tensor '
                                                  'groups) + its step form, nothing else changed: the drop-in training call.  The optimizer object is bound to the fused step at its first '
                                                  'step() (invr.optim.adopt_on_first_step, installed by NetworkWrapper: shared param_groups / state, gradient arena); '
                                                  'INVR_NO_OPTIM_HOOK=1 leaves torch\'s own Adam (9.3 ms in round 4).  train_step = the same with driver.make_optimizer (FusedAdam)')
            except Exception as e:
                line['api_train_step'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(net, cfg, batch_cpu, min(args.cpu_rays, n_rays), S)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
