#!/usr/bin/env python
"""Stress of the frames-in-flight modes: K frames rendered side by side must be, replay after replay, bit for bit what
K separate renders give.  On the first differing element the tool says which frame / tensor / element, then walks the frame's workspace
in pipeline order against a snapshot taken after a clean replay and names the first array (= kernel stage) that diverged.

  python tools/stress_replay.py --config small --mode graph --replays 2000      (the shape the one-off mismatch of round 5 was seen at)
  python tools/stress_replay.py --config bench --mode graph --replays 300      (10 branches, 512x512x128, full tables)
  python tools/stress_replay.py --config bench --mode lanes --replays 300      (Renderer.in_flight = 8)
  --rebuild N : tear the FrameSet down and capture it again every N replays (allocator / capture states)
Prints one JSON line: {"config", "mode", "replays", "mismatching_replays", "first": {...}}; exit code 1 on any mismatch."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ctypes as C_                  # noqa: E402
import torch                         # noqa: E402

import invr                          # noqa: E402,F401
from invr import _abi                # noqa: E402
from invr import frames as iframes   # noqa: E402
import bench                         # noqa: E402

KEYS = ('rgb_map', 'acc_map', 'raw', 'occ')
# workspace arrays in the order the pipeline writes them (the first one that differs names the stage)
STAGES = [('cull/compact', ('mask', 'byte_off', 'word_off', 'active_idx')), ('knn', ('pflags', 'farflags', 'l_nn', 'l_w')), ('pair_lists', ('l_slot', 'counters')),
          ('warp+deform', ('l_x', 'l_d', 'l_r')), ('encode', ('emb',)), ('occ mlp', ('occp',)), ('winner lists', ('wsel', 'wcnt', 'wl')), ('rgb mlp', ('rgbw',))]


def refs_of(net, batches, S, rank=0, world=1):
    from invr import dist as idist
    refs = []
    for b in batches:
        dev = b['ray_o'].device
        idx = idist.tile_indices(b['ray_o'].shape[1], rank, world, device=dev)
        net._ws = None
        o = net.render_rays(b, b['ray_o'][0][idx], b['ray_d'][0][idx], b['near'][0][idx], b['far'][0][idx], S, want_raw=True)
        refs.append({k: o[k].clone() for k in KEYS})
    net._ws = None
    torch.cuda.synchronize()
    return refs


def first_diff(a, b):
    ne = (a != b) if a.dtype != torch.float32 else (a.view(torch.int32) != b.view(torch.int32))
    n = int(ne.sum())
    if not n:
        return None
    flat = ne.reshape(-1).nonzero()[:8, 0].tolist()
    d = {'count': n, 'of': a.numel(), 'first_flat_indices': flat}
    if a.dtype == torch.float32:
        d['max_abs'] = float((a.double() - b.double()).abs().nan_to_num(nan=1e30).max())
        d['got_vs_ref'] = [(float(a.reshape(-1)[i]), float(b.reshape(-1)[i])) for i in flat[:4]]
    return d


def ws_diff(ws_now, ws_good, wsinfo, stats):
    """first workspace array (pipeline order) whose USED part differs from the clean snapshot"""
    _, n, S, cap = wsinfo
    vn, vg = _abi.ws_views(ws_now, n, S, cap), _abi.ws_views(ws_good, n, S, cap)
    na, pairs = int(stats[0]), [int(x) for x in stats[1:6]]
    out = []
    for stage, names in STAGES:
        for name in names:
            a, b = vn[name], vg[name]
            items = list(zip(a, b)) if isinstance(a, list) else [(a, b)]
            for p, (x, y) in enumerate(items):
                if name in ('l_nn', 'l_w', 'pflags', 'farflags', 'wsel', 'active_idx'):
                    x, y = x[:na], y[:na]          # indexed by survivor slot
                elif name in ('l_slot', 'occp', 'wl') and isinstance(a, list):
                    x, y = x[:pairs[p]], y[:pairs[p]]
                elif name in ('l_x', 'l_d', 'l_r', 'emb'):
                    x, y = x[:, :pairs[p]], y[:, :pairs[p]]
                elif name == 'rgbw':
                    x, y = x[:na], y[:na]
                elif name == 'counters':
                    x, y = x[:7], y[:7]
                d = first_diff(x.contiguous(), y.contiguous())
                if d:
                    # which list entries / slots (the LAST axis of the array) differ: later stages inherit an earlier stage's entries;
                    # an entry set of its own means that stage was hit independently
                    xe, ye = x.contiguous(), y.contiguous()
                    ne = (xe.view(torch.int32) != ye.view(torch.int32)) if xe.dtype == torch.float32 else (xe != ye)
                    if ne.dim() == 2 and name in ('l_x', 'l_d', 'l_r', 'emb'):
                        ent = ne.any(0).nonzero()[:, 0]
                    elif ne.dim() == 2:
                        ent = ne.any(1).nonzero()[:, 0]
                    else:
                        ent = ne.nonzero()[:, 0]
                    d['entries'] = ent[:48].tolist()
                    d['n_entries'] = int(ent.numel())
                    if name in ('l_x', 'l_d') and hasattr(_abi.lib(), 'invr_debug_warp_dump_offset') and ent.numel():
                        # -DWARP_DUMP build: k_warp_pairs left its intermediates [part][36][lcap] in the last array of the workspace:
                        # rows 0-11 A_bw, 12-23 B_bw, 24-26 pose point, 27-29 pose direction, 30-32 x_b, 33-35 d_b
                        L_ = _abi.lib()
                        L_.invr_workspace_bytes.restype = C_.c_size_t
                        tot = int(L_.invr_workspace_bytes(C_.c_int64(n), C_.c_int32(S), C_.c_int64(cap)))
                        lc = vn['lcap']
                        size = lc * 5 * 36 * 4
                        off = (tot - size) // 256 * 256
                        dn = ws_now[off:off + size].view(torch.float32).view(5, 36, lc)[p][:, ent[:16]]
                        dg = ws_good[off:off + size].view(torch.float32).view(5, 36, lc)[p][:, ent[:16]]
                        rows = (dn.view(torch.int32) != dg.view(torch.int32)).any(1).nonzero()[:, 0].tolist()
                        d['intermediates_differing_rows'] = rows
                        d['intermediates'] = {str(r): {'now': dn[r][:6].tolist(), 'good': dg[r][:6].tolist()} for r in rows[:12]}
                    if name in ('l_x', 'l_d') and ent.numel():
                        e = ent[:16]
                        d['dump'] = {'l_x_now': vn['l_x'][p][:, e].tolist(), 'l_x_good': vg['l_x'][p][:, e].tolist(),
                                     'l_d_now': vn['l_d'][p][:, e].tolist(), 'l_d_good': vg['l_d'][p][:, e].tolist(),
                                     'prev_iter_l_d_good': vg['l_d'][p][:, (e - 1024 * 128).clamp(min=0)].tolist(),
                                     'next_iter_l_d_good': vg['l_d'][p][:, (e + 1024 * 128).clamp(max=pairs[p] - 1)].tolist(),
                                     'slot': vn['l_slot'][p][e].tolist()}
                    out.append({'stage': stage, 'array': name, 'part': p if isinstance(a, list) else None, **d})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', choices=['small', 'bench', 'shard'], default='small')
    ap.add_argument('--mode', choices=['graph', 'streams', 'lanes', 'eager'], default='graph')
    ap.add_argument('--replays', type=int, default=200)
    ap.add_argument('--rebuild', type=int, default=0)
    ap.add_argument('--frames', type=int, default=0)
    ap.add_argument('--check-every', type=int, default=1, help='compare after every Nth replay (the replays in between run back to back)')
    ap.add_argument('--snapshot', action='store_true', help='keep a clean copy of every frame\'s workspace to name the diverging stage')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    if args.config == 'small':
        from invr import params, scene
        from invr.config import make_cfg
        from invr.network import Network
        cfg = make_cfg(table_log2=12, N_samples=64)
        net = Network(cfg=cfg)
        net.load_state_dict(params.init_state_dict(cfg, seed=4), strict=True)
        net = net.to(dev).eval()
        K, S, world = args.frames or 4, 64, 1
        batches = []
        for k in range(K):
            b, _ = scene.make_scene(128, 128, seed=0, cam_dist=1.8, frame=3 + 7 * k, pose_seed=k)
            batches.append({kk: v.to(dev) for kk, v in scene.to_torch(b).items()})
    else:
        from invr.config import make_cfg
        cfg = make_cfg(N_samples=128)
        cfg['eval_row_sums'] = True
        net = bench.build_model(cfg, dev)
        K, S = args.frames or 10, 128
        world = 8 if args.config == 'shard' else 1
        _, batches = bench.frame_batches(512, 1.8, K, dev)
    refs = refs_of(net, batches, S, 0, world)
    bad, first, checked = 0, None, 0
    import ctypes as C
    L = _abi.lib()
    have_dbg = hasattr(L, 'invr_debug_warp')           # the -DWARP_VERIFY variant (INVR_LIB_PATH=variants/libinvr_verify.so)
    dbg_events = []

    def read_dbg(rep):
        if not have_dbg:
            return
        buf = (C.c_ulonglong * (8 + 64 * 8))()
        L.invr_debug_warp(buf, 1)
        if buf[1]:
            recs = []
            for n in range(min(int(buf[1]), 64)):
                r = buf[8 + n * 8: 16 + n * 8]
                import struct
                f = lambda b: struct.unpack('f', struct.pack('I', b & 0xFFFFFFFF))[0]
                recs.append({'p': r[0], 'i': r[1], 'lane': r[1] % 64, 'c': r[2] & 255, 'arr': ('l_x', 'l_d', 'reload of vmat row (c = neighbour * 6 + row; got = first load, expect = second)', 'blend FMAs twice (got = first, expect = second)', 'pose point twice', 'warp_with_mats twice', 'slot / nn / weights loaded twice')[min(int(r[2] >> 8), 6)], 'got': f(r[3]), 'expect': f(r[4]),
                             'hw_id': hex(r[5]), 'xcc': r[6] & 15, 'block': r[7]})
            dbg_events.append({'replay': rep, 'verify_launches': int(buf[0]), 'differing': int(buf[1]), 'records': recs})
            sys.stderr.write('VERIFY replay %d: %d differing values right behind k_warp_pairs: %s\n' % (rep, buf[1], json.dumps(recs[:20])))
    if have_dbg:
        L.invr_debug_warp(None, 1)

    def compare(get, rep, wsinfo=None):
        nonlocal bad, first
        miss = []
        for k in range(K):
            out = get(k)
            for key in KEYS:
                a = out[key].reshape(refs[k][key].shape) if out[key].shape != refs[k][key].shape else out[key]
                if not torch.equal(a, refs[k][key]):
                    miss.append((k, key, first_diff(a.contiguous(), refs[k][key])))
        if miss:
            bad += 1
            if first is None:
                first = {'replay': rep, 'tensors': [{'frame': k, 'key': key, **(d or {})} for k, key, d in miss]}
                if wsinfo is not None:
                    k0 = miss[0][0]
                    info, good, stats = wsinfo(k0)
                    if good is not None:
                        first['workspace'] = ws_diff(info[0], good, info, stats)
                sys.stderr.write('MISMATCH at replay %d: %s\n' % (rep, json.dumps(first)[:3000]))
        return not miss

    if args.mode in ('graph', 'streams', 'eager'):
        fs = None
        good = None
        for rep in range(args.replays):
            if fs is None or (args.rebuild and rep % args.rebuild == 0):
                fs = None
                torch.cuda.empty_cache()
                fns, n_rays, keep = iframes.shard_render_fns(net, batches, S, 0, world, want_raw=True)
                fs = iframes.FrameSet(fns, n_rays, device=dev, capture=args.mode == 'graph', streams=args.mode == 'streams')
                good = None
            fs.replay()
            if (rep + 1) % args.check_every:
                continue
            torch.cuda.synchronize()
            checked += 1
            read_dbg(rep)
            assert iframes.check_overflow(fs)

            def wsinfo(k):
                return fs.local[k]['_ws'], (good[k] if good else None), fs.local[k]['stats'].cpu()
            ok = compare(lambda k: fs.local[k], rep, wsinfo)
            if ok and args.snapshot and good is None:
                good = [fs.local[k]['_ws'][0].clone() for k in range(K)]
    else:
        from collections import deque
        from invr.renderer import Renderer
        r = Renderer(net)
        r.in_flight, r.eval_to_cpu = 8, False
        q = deque()
        rep = 0

        def check_one(o, k, rep):
            nonlocal checked
            checked += 1
            got = {key: o[key][0] for key in KEYS}
            got['occ'] = got['occ'][:, 0]
            compare_one = lambda kk: got
            miss = [(key, first_diff(got[key].contiguous(), refs[k][key])) for key in KEYS if not torch.equal(got[key], refs[k][key])]
            nonlocal bad, first
            if miss:
                bad += 1
                if first is None:
                    first = {'replay': rep, 'tensors': [{'frame': k, 'key': key, **(d or {})} for key, d in miss]}
                    sys.stderr.write('MISMATCH at frame %d: %s\n' % (rep, json.dumps(first)[:3000]))
        for rep in range(args.replays * K):
            k = rep % K
            q.append((r.render(dict(batches[k])), k, rep))
            while len(q) >= r.in_flight:
                o, kk, rr = q.popleft()
                check_one(o, kk, rr)
        while q:
            o, kk, rr = q.popleft()
            check_one(o, kk, rr)
        r.flush(release=True)
        read_dbg(-1)
    print(json.dumps({'config': args.config, 'mode': args.mode, 'frames_in_flight': K if args.mode != 'lanes' else 8, 'replays': args.replays,
                      'checked': checked, 'mismatching': bad, 'first': first, 'verify_events': dbg_events[:20]}))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
