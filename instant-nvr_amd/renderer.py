"""Drop-in for lib/networks/renderer/inb_renderer.py::Renderer (reference :11-239).

Same constructor and ``render(batch, test=False, epoch=-1)`` signature and the same output
dict; the per-chunk Python loop, ``get_wsampling_points``, ``get_density_color``, the network
forward and ``volume_rendering`` are one stream-ordered libinvr call over the whole ray list
(chunk-free: HBM holds the full frame's intermediates).

Train mode (``net.training`` and grad enabled) goes through autograd.TrainRenderFn — one fused HIP forward and one fused
HIP backward — and returns the reference's training dict (rgb_map, acc_map, weights / z_vals, resd, tpts, tocc, oresd,
distortion) with the large per-pair tensors materialised lazily (LazyTrainRet) plus two scalars the trainer prefers when
present: ``offset_loss`` and ``pair_loss`` (the reference's means over resd / oresd, reduced on the device).  ``tpts`` is the
reference's ``init_bigpose`` for every one of the Na x 5 (survivor, part) rows, flagged or not
(inb_part_network_multiassign.py:96-120,162-166): rows the pair pruning never listed are filled through the dense stage entry
points when the tensor is materialised (``dense_train_rows``).
"""
import ctypes as C

import torch

from . import _abi
from .config import cfg as global_cfg, NUM_PARTS

MAX_SAMPLES_PER_CALL = (1 << 31) - 1


def dense_train_rows(v, stats, ctx, rays, S, jitter):
    """The reference's dense train-mode tensors resd / tpts / tocc, (Na, P, .) in its row order
    (inb_part_network_multiassign.py:96-120,162-166), from what a forward left in the workspace views `v`:
      * listed pairs: the pair lists (l_r = resd, l_x - l_r = init_bigpose, occp = tocc),
      * far pairs: the part's far constant (last list entry; the reference's value differs from it by < 1.1e-9 m, k_knn.hip),
      * every other (survivor, part): resd = 0 and tocc = 0 as in the reference (deformer / part network skip unflagged rows),
        tpts = init_bigpose under that part's 4-NN weights — the reference warps all Na x P rows — through the dense stage
        entry points invr_pose_points -> invr_knn_blend -> invr_warp_deform (Na is a training patch's ~4e4 survivors).
    `stats` is the host copy of the statistics block, `rays` = (ray_o, ray_d, near, far) of the forward, `jitter` its z jitter."""
    from . import stages
    Na, cap, P = int(stats[0]), v['cap'], NUM_PARTS
    dev = v['pflags'].device
    resd = torch.zeros(Na + 1, P, 3, device=dev)
    tpts = torch.zeros(Na + 1, P, 3, device=dev)
    tocc = torch.zeros(Na + 1, P, device=dev)
    far = v['farflags'][:Na].to(torch.int32)
    for p in range(P):
        cnt = int(stats[1 + p])
        slots = v['l_slot'][p][:cnt].long()
        rows = torch.where(slots == cap, torch.full_like(slots, Na), slots)       # const pair -> extra row
        r = v['l_r'][p][:, :cnt].t()
        resd[rows, p] = r
        tpts[rows, p] = v['l_x'][p][:, :cnt].t() - r                                 # init_bigpose
        tocc[rows, p] = v['occp'][p][:cnt]
        fr = ((far >> p) & 1).nonzero(as_tuple=True)[0]                              # far pairs take the constant
        if fr.numel():
            resd[fr, p] = resd[Na, p]
            tpts[fr, p] = tpts[Na, p]
            tocc[fr, p] = tocc[Na, p]
    resd, tpts, tocc = resd[:Na], tpts[:Na], tocc[:Na]
    if Na:
        ro, rd, nr, fa = rays
        pts, dirs = stages.pose_points(ctx.scene, ro, rd, nr, fa, S, v['active_idx'][:Na], jitter=jitter)
        bw, _ = stages.knn_blend(ctx.scene, pts)
        tp, _, _ = stages.warp_deform(ctx.scene, ctx.model, pts, dirs, bw, torch.zeros(Na, P, dtype=torch.uint8, device=dev))
        known = (((v['pflags'][:Na].to(torch.int32) | far)[:, None] >> torch.arange(P, device=dev)[None]) & 1).bool()
        tpts = torch.where(known[..., None], tpts, tp)
    return resd, tpts, tocc


class LazyHostRet(dict):
    """The eval return dict of Renderer.render under the reference's move-everything-to-the-host contract (inb_renderer.py:199-200).
    The image maps (rgb_map, acc_map: what the reference's evaluator and visualizers read, evaluators/if_nerf.py:77,
    visualizers/if_nerf.py:24) are host tensors when render() returns; the per-sample tensors (raw, occ: 656 MB for a 512x512x128
    frame, 12 ms of PCIe for a 2.5 ms render) are copied on first access — by key, or all of them by keys() / items() / values() /
    iteration — and are host tensors of the reference's shapes from then on.  The device tensors they come from stay alive inside
    this object until then; a dict built from it (`dict(ret)`) is the plain all-host dict."""
    def __init__(self, host, lazy_dev, pin, pending=None, keys=()):
        super().__init__(host)
        self._lazy, self._pin = dict(lazy_dev), pin
        # a frame still in flight on one of the renderer's lanes (Renderer.in_flight > 1): `pending.result()` joins it — a host wait
        # on THAT frame's event only — and hands over (host maps, device tensors for the lazy keys); `keys` = the keys it will have
        self._pending, self._keys = pending, tuple(keys)

    def _settle(self):
        if self._pending is not None:
            pend, self._pending = self._pending, None
            host, lazy = pend.result()
            dict.update(self, host)
            self._lazy = dict(lazy)

    def _fetch(self, keys):
        self._settle()
        todo = [k for k in keys if k in self._lazy]
        if not todo:
            return
        on_dev = False
        for k in todo:
            v = self._lazy.pop(k).detach()
            h = torch.empty(v.shape, dtype=v.dtype, device='cpu', pin_memory=self._pin and v.is_cuda)
            h.copy_(v, non_blocking=True)
            on_dev = on_dev or v.is_cuda
            dict.__setitem__(self, k, h)
        if on_dev:
            torch.cuda.current_stream().synchronize()

    def _fetch_all(self):
        # join the frame FIRST: while it is in flight `_lazy` is still empty, and tuple(self._lazy) taken before the join would fetch nothing
        self._settle()
        self._fetch(tuple(self._lazy))

    def __contains__(self, k):
        if self._pending is not None:
            return k in self._keys
        return dict.__contains__(self, k) or k in self._lazy

    def __getitem__(self, k):
        self._fetch((k,))
        return dict.__getitem__(self, k)

    def get(self, k, d=None):
        return self[k] if k in self else d

    def __len__(self):
        if self._pending is not None:
            return len(self._keys)
        return dict.__len__(self) + len(self._lazy)

    def __iter__(self):
        self._fetch_all()
        return dict.__iter__(self)

    def keys(self):
        self._fetch_all()
        return dict.keys(self)

    def items(self):
        self._fetch_all()
        return dict.items(self)

    def values(self):
        self._fetch_all()
        return dict.values(self)

    def pop(self, k, *default):
        self._fetch((k,))
        return dict.pop(self, k, *default)

    def copy(self):
        self._fetch_all()
        return dict(dict.items(self))           # (the plain all-host dict)

    def __eq__(self, other):
        self._fetch_all()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        if self._pending is not None:
            return 'LazyHostRet(<frame in flight: %s>)' % ', '.join(self._keys)
        return 'LazyHostRet(%s%s)' % (dict.__repr__(self), ''.join(', %s: <on the device>' % k for k in self._lazy))

    def __reduce__(self):                       # pickle / copy.deepcopy: the plain all-host dict
        self._fetch_all()
        return (dict, (dict(dict.items(self)),))

    def pending(self):
        """keys whose host copy has not been made yet"""
        if self._pending is not None:
            return tuple(self._keys)
        return tuple(self._lazy)

    def device_bytes(self):
        """bytes of device memory the not-yet-fetched tensors of a joined frame keep alive (occ is a view of raw: counted once)"""
        if self._pending is not None:
            return 0
        seen = {}
        for v in self._lazy.values():
            if v.is_cuda:
                st = v.untyped_storage()
                seen[st.data_ptr()] = st.nbytes()
        return sum(seen.values())

    def fetch(self):
        """make every entry a host tensor now (releases the device tensors)"""
        self._fetch_all()

    def in_flight(self):
        """True while the frame behind this dict has not been joined (Renderer.in_flight > 1)"""
        return self._pending is not None and not self._pending.done()


class LazyDevRet(dict):
    """The eval return dict of a frame in flight when the outputs stay on the device (Renderer.eval_to_cpu = False with
    Renderer.in_flight > 1): every access joins the frame first (a host wait on that frame's event, which also checks its workspace
    for overflow); from then on it is the plain dict of device tensors."""
    def __init__(self, pending, keys):
        super().__init__()
        self._pending, self._keys = pending, tuple(keys)

    def _settle(self):
        if self._pending is not None:
            pend, self._pending = self._pending, None
            dev, _ = pend.result()
            dict.update(self, dev)

    def __contains__(self, k):
        return k in self._keys if self._pending is not None else dict.__contains__(self, k)

    def __len__(self):
        return len(self._keys) if self._pending is not None else dict.__len__(self)

    def __getitem__(self, k):
        self._settle()
        return dict.__getitem__(self, k)

    def get(self, k, d=None):
        self._settle()
        return dict.get(self, k, d)

    def __iter__(self):
        self._settle()
        return dict.__iter__(self)

    def keys(self):
        self._settle()
        return dict.keys(self)

    def items(self):
        self._settle()
        return dict.items(self)

    def values(self):
        self._settle()
        return dict.values(self)

    def pop(self, k, *default):
        self._settle()
        return dict.pop(self, k, *default)

    def copy(self):
        self._settle()
        return dict(self)

    def __eq__(self, other):
        self._settle()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        return 'LazyDevRet(<frame in flight: %s>)' % ', '.join(self._keys) if self._pending is not None else dict.__repr__(self)

    def __reduce__(self):
        self._settle()
        return (dict, (dict(self),))

    def in_flight(self):
        return self._pending is not None and not self._pending.done()


class _Lane:
    """One of the renderer's K frame slots: a stream, the workspace of the frames rendered there (only ever touched on that stream,
    so frame f + K queues behind frame f and nothing else) and the frame in flight on it."""
    def __init__(self, device):
        self.stream = torch.cuda.Stream(device)
        self.ws = None
        self.pending = None
        self.stats_host = None       # a page-locked copy target for the frame's statistics block (64 bytes, allocated once per lane)
        self.raw_buf = None          # the lane's `raw` buffer (0.5 GB for 512x512x128): handed to the next frame once nobody references it

    def raw_buffer(self, numel, device):
        """A flat float32 buffer of >= numel elements for this frame's raw: the lane's previous one when every dict / tensor that
        viewed it is gone (the common loop reads the maps and drops the dict: no 0.5 GB allocation per frame — hipMalloc of such
        blocks took up to seconds when the caching allocator had to go back to the driver), else a fresh one."""
        buf = self.raw_buf
        if buf is not None and buf.numel() >= numel and buf.device == device:
            try:
                free = torch._C._storage_Use_Count(buf.untyped_storage()._cdata) <= 2          # `buf` itself + the wrapper just made
            except Exception:
                free = False
            if free:
                return buf
        self.raw_buf = torch.empty(int(numel * 1.05) + 1024, device=device, dtype=torch.float32)
        return self.raw_buf

    def stats_buffer(self, like):
        if self.stats_host is None or self.stats_host.shape != like.shape:
            self.stats_host = torch.empty(like.shape, dtype=like.dtype, pin_memory=True)
        return self.stats_host


class _PendingFrame:
    """An eval frame launched on a lane.  `result()` joins it (event wait), reads its statistics block, re-renders it at full
    workspace capacity if the survivor bound was too small (stats[6]: nothing was written out of bounds, the frame is just
    incomplete), and returns (maps, lazy): eval_to_cpu -> host maps + device raw / occ for LazyHostRet, else the device dict."""
    def __init__(self, renderer, lane, call, cap, keep):
        self.r, self.lane, self.call, self.cap, self.keep = renderer, lane, call, cap, keep
        self.out = self.stats_host = self.event = None
        self._res = None

    def launch(self):
        self.out = self._run(self.cap)

    def _run(self, cap):
        """the frame on the lane's stream; every device tensor comes from the CALLER's stream pool (Network.render_rays(stream=))"""
        r, lane, net = self.r, self.lane, self.r.net
        dev = lane.stream.device
        lane.stream.wait_stream(torch.cuda.current_stream(dev))          # inputs + freshly allocated blocks are ordered on the caller's stream
        prev, net._ws = net._ws, lane.ws
        try:
            out = self.call(cap, lane.stream)
        finally:
            lane.ws, net._ws = net._ws, prev
        with torch.cuda.stream(lane.stream):
            # only the 64-byte statistics block follows the render asynchronously (the lane's own page-locked buffer: the frame that
            # used it last was joined before the lane rendered again).  The image maps are copied when the frame is JOINED
            # (result()): into ordinary host tensors — per-frame page-locked allocations stalled for 25 .. 90 ms in some allocator
            # states, and page-locked memory is uncached for the CPU here (a consumer reads a 3 MB map in 4.6 ms instead of 0.03)
            self.stats_host = lane.stats_buffer(out['stats'])
            self.stats_host.copy_(out['stats'], non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(lane.stream)
        return out

    def done(self):
        return self._res is not None or self.event.query()

    def result(self):
        if self._res is not None:
            return self._res
        r, lane = self.r, self.lane
        self.event.synchronize()
        redo = bool(self.cap) and int(self.stats_host[6]) != 0
        if redo:                                                          # survivor bound too small: once more at full capacity
            lane.ws = None                                                # (not both workspaces at once)
            self.out = self._run(0)
            lane.stream.synchronize()
            lane.ws = None          # ~1150 B per ray-sample: the next frame of this lane gets one sized from the new survivor count
        st = self.stats_host.clone()                                      # (the lane's buffer is reused by its next frame)
        if int(st[6]) != 0:
            raise RuntimeError('invr_render_fwd reported stats[6] = %d: workspace overflow' % int(st[6]))
        # grow-only: the K lanes' workspaces settle at ONE size — the largest frame seen — after a few frames.  (Following every
        # frame's own count made a lane reallocate whenever a denser frame than any before landed on it: a multi-GB hipMalloc,
        # ~1 s each on this runtime, for the first LCM(K, sequence period) frames.)
        # (1.25 x the largest frame seen; the FIRST frame of a sequence adds its own head-room in _render_in_flight.  Round 5 took 1.5 x
        # here and 1.25 x on top of it there: 1.9 x the survivors, 6.5 GB per lane)
        r._cap_hint = max(r._cap_hint or 0, int(1.25 * int(st[0])) + 65536)
        r.last_stats = self.out['stats']
        out = self.out
        cur = torch.cuda.current_stream(out['rgb_map'].device)
        dev = {'rgb_map': out['rgb_map'][None], 'acc_map': out['acc_map'][None]}
        if r.want_raw:
            dev['raw'] = out['raw'][None]
            dev['occ'] = out['occ'][None, :, None]
        for v in dev.values():
            v.record_stream(cur)               # (rendered on the lane's stream, read by the caller on its own)
        if lane.pending is self:
            lane.pending = None
        self.keep = self.call = None           # the frame's inputs may go
        if r.eval_to_cpu:
            lazy = {k: dev[k] for k in ('raw', 'occ') if k in dev}
            maps = LazyHostRet({}, {k: dev[k] for k in ('rgb_map', 'acc_map')}, r.pin_host)          # 4 MB, now: the frame is complete, the
            maps._fetch(('rgb_map', 'acc_map'))                                                       # other lanes keep rendering beside the copy
            host = dict(maps)
            if not r.lazy_host:
                tmp = LazyHostRet(host, lazy, r.pin_host)
                tmp._fetch(tuple(lazy))
                host, lazy = dict(tmp), {}
            self._res = (host, lazy)
        else:
            self._res = (dev, {})
        self.out = None
        return self._res


class Renderer:
    def __init__(self, net):
        self.net = net
        self.cfg = getattr(net, 'cfg', global_cfg)
        # knobs that are not part of the reference signature
        self.eval_to_cpu = True        # reference moves every eval output to the CPU (:199-200)
        self.want_raw = True           # reference always returns raw/occ
        self.adaptive_cap = True       # size the workspace from the previous frame's survivor count (eval_to_cpu only)
        self.pin_host = False          # eval_to_cpu: True = page-locked host tensors for the outputs (a faster copy: 26 instead of ~100 ms for
                                       # the 656 MB of raw + occ) — off by default since round 5: torch's page-locked memory is UNCACHED for
                                       # the CPU on this platform, a consumer reads it at ~1 GB/s (4.6 ms per pass over a 3 MB rgb_map,
                                       # 0.03 ms from an ordinary tensor), which costs a caller more than the copy saved
        self.lazy_host = True          # eval_to_cpu: raw / occ reach the host on first access (LazyHostRet); False: with the maps
        # Frames in flight across render() calls (cfg.render_in_flight, default 1 = every call on the caller's stream as before).
        # K > 1: eval frame f is launched on lane f % K — a stream and a workspace of its own — and render() returns at once with a
        # dict that joins the frame on first access (LazyHostRet / LazyDevRet): a caller that reads frame f after submitting frames
        # f + 1 .. f + K - 1 (driver.run_evaluate) keeps K kernel chains on the GPU, which fill each other's ramps and tails
        # (measured, 512x512x128: 2.37 / 2.07 / 2.04 / 1.94 ms per frame for K = 1 / 2 / 4 / 8, eager launches, no graph — the
        # batches of a sequence differ in their volume dimensions, so there is nothing static to capture).  A caller that reads
        # every frame at once (run.py:61-90 as written) gets the latency of one frame either way.
        self.in_flight = int(self.cfg.get('render_in_flight', 1)) if hasattr(self.cfg, 'get') else 1
        self._lanes, self._lane_i = [], 0
        # raw / occ of returned dicts stay on the device until somebody reads them (LazyHostRet): a caller that COLLECTS the dicts of
        # a sequence and never reads raw would pin 656 MB of HBM per 512x512x128 frame.  At most this many bytes stay device-side
        # behind returned dicts; beyond it the oldest dicts are completed on the host (what the reference does for every frame)
        self.lazy_device_budget = int(self.cfg.get('lazy_device_budget', 8 << 30)) if hasattr(self.cfg, 'get') else 8 << 30
        self._lazy_live = []
        self._cap_hint = None

    def render(self, batch, test=False, epoch=-1):
        cfg = self.cfg
        ray_o, ray_d = batch['ray_o'], batch['ray_d']
        near, far = batch['near'], batch['far']
        if epoch != -1:
            batch['epoch'] = epoch
        n_batch, n_pixel = ray_o.shape[:2]
        assert n_batch == 1, 'the path asserts a batch of one frame (inb_part_network_multiassign.py:84,155)'
        S = int(cfg.N_samples)
        training = self.net.training
        jitter = None
        if cfg.perturb > 0. and training:
            jitter = self._jitter((n_pixel, S), ray_o.device)                              # :24
        if training:
            return self._render_train(batch, jitter)
        per_call = max(1, MAX_SAMPLES_PER_CALL // S)
        if self.in_flight > 1 and ray_o.is_cuda and n_pixel <= per_call:
            return self._render_in_flight(batch, ray_o[0], ray_d[0], near[0], far[0], S, jitter)
        outs = []
        for i in range(0, n_pixel, per_call):
            sl = slice(i, i + per_call)
            n_samp = min(per_call, n_pixel - i) * S
            # The workspace is ~850 B per possible survivor.  With max_active = 0 every ray-sample may survive (28 GB
            # for 512x512x128); since the eval contract ends in a host copy anyway, the survivor capacity follows the
            # previous frame (x1.5) and a frame that overflows it (stats[6]) is rendered again at full capacity.
            cap = 0
            if self.adaptive_cap and self.eval_to_cpu:
                cap = min(n_samp, max(self._cap_hint if self._cap_hint is not None else n_samp // 4, 65536))
            call = lambda c: self.net.render_rays(batch, ray_o[0, sl], ray_d[0, sl], near[0, sl], far[0, sl], S,
                                                  jitter=None if jitter is None else jitter[sl], want_raw=self.want_raw, max_active=c)
            out = call(cap)
            if cap:
                st = out['stats'].cpu()
                if int(st[6]) != 0:
                    out = call(0)
                    st = out['stats'].cpu()
                self._cap_hint = max(self._cap_hint or 0, int(1.5 * int(st[0])) + 65536)          # grow-only (see _PendingFrame.result)
            elif self.eval_to_cpu:
                st = out['stats'].cpu()
            else:
                st = None                                   # asynchronous caller: check last_stats[6] yourself
            if st is not None and int(st[6]) != 0:
                raise RuntimeError('invr_render_fwd reported stats[6] = %d: workspace overflow' % int(st[6]))
            outs.append(out)
        cat = (lambda k: outs[0][k]) if len(outs) == 1 else (lambda k: torch.cat([o[k] for o in outs], 0))
        ret = {'rgb_map': cat('rgb_map')[None], 'acc_map': cat('acc_map')[None]}
        if self.want_raw:
            ret['raw'] = cat('raw')[None]
            ret['occ'] = cat('occ')[None, :, None]
        self.last_stats = outs[-1]['stats']
        if self.eval_to_cpu:
            # the reference moves every eval output to the host (:199-200).  raw + occ of a 512x512x128 frame are 656 MB: through
            # pageable memory that copy takes ~25x the render; page-locked destinations (torch's caching host allocator: fresh
            # tensors per call, no aliasing between frames) and one stream synchronisation bring it to the PCIe rate
            # — for the image maps at once, for raw / occ on first access (LazyHostRet; lazy_host = False: all four at once)
            lazy = {k: ret[k] for k in ('raw', 'occ') if self.lazy_host and k in ret}
            host = LazyHostRet({}, {k: v for k, v in ret.items() if k not in lazy}, self.pin_host)
            host._fetch(tuple(host._lazy))
            host._lazy = lazy
            ret = self._track_lazy(host) if lazy else host
        return ret

    def _render_in_flight(self, batch, ray_o, ray_d, near, far, S, jitter):
        """One eval frame on the next lane (see __init__); returns a dict that joins the frame on first access."""
        dev = ray_o.device
        K = self.in_flight
        while len(self._lanes) < K:
            self._lanes.append(_Lane(dev))
        lane = self._lanes[self._lane_i % K]
        self._lane_i += 1
        if lane.pending is not None:
            lane.pending.result()              # the frame rendered here K calls ago (long done): releases its inputs, updates the cap hint
        n_samp = ray_o.shape[0] * S
        cap = min(n_samp, max(self._cap_hint if self._cap_hint is not None else n_samp // 4, 65536)) if self.adaptive_cap else 0
        if lane.ws is not None and lane.ws.numel() < _abi.lib().invr_workspace_bytes(ray_o.shape[0], S, cap):
            # the survivor bound grew past this lane's workspace: give the old block back to the DRIVER before the larger one is
            # allocated — the caching allocator would keep it (a 4 GB block nobody can reuse, per lane and regrowth: 176 GB reserved
            # for 8 lanes in round 5).  Rare (the bound is grow-only), and the regrowth costs a multi-GB hipMalloc anyway.
            lane.ws = None
            torch.cuda.empty_cache()
        self._raw_numel = max(getattr(self, '_raw_numel', 0), n_samp * 4)          # grow-only, renderer-wide: every lane's raw buffer fits the largest frame seen
        ctx = self.net.prepare(batch)          # on the caller's stream: a stale row-sum table is rebuilt in front of every lane
        call = lambda c, st: self.net.render_rays(ctx, ray_o, ray_d, near, far, S, jitter=jitter, want_raw=self.want_raw, max_active=c, stream=st,
                                                  raw_out=lane.raw_buffer(self._raw_numel, dev) if self.want_raw else None)
        pend = _PendingFrame(self, lane, call, cap, (batch, ctx, ray_o, ray_d, near, far, jitter))
        pend.launch()
        lane.pending = pend
        if self._cap_hint is None and self.adaptive_cap:
            # the first frame of a sequence is joined at once: its survivor count sizes the workspaces of all later frames (K lanes at
            # the no-hint default of a quarter of the ray-samples would be K x 9.6 GB for 512x512x128)
            pend.result()
            if lane.ws is not None and self._cap_hint is not None and cap > 2 * self._cap_hint:
                lane.ws = None
            if self._cap_hint is not None:
                self._cap_hint = int(1.2 * self._cap_hint)           # head-room over the first frame (1.5 x its survivors in all): later frames of a sequence rarely force a regrowth
        keys = ('rgb_map', 'acc_map') + (('raw', 'occ') if self.want_raw else ())
        if self.eval_to_cpu:
            return self._track_lazy(LazyHostRet({}, {}, self.pin_host, pending=pend, keys=keys))
        return LazyDevRet(pend, keys)

    def _track_lazy(self, ret):
        """bound the device memory behind returned LazyHostRet dicts (lazy_device_budget); called with every new dict"""
        import weakref
        self._lazy_live.append(weakref.ref(ret))
        live, total = [], 0
        for w in self._lazy_live:
            r = w()
            if r is not None and (r._pending is not None or r._lazy):
                live.append(w)
                total += r.device_bytes()
        self._lazy_live = live
        for w in live:                                   # oldest first
            if total <= self.lazy_device_budget:
                break
            r = w()
            if r is not None and r._pending is None:
                total -= r.device_bytes()
                r.fetch()
        return ret

    def flush(self, release=False):
        """Join every frame in flight (their dicts stay valid); release=True also gives the lanes' workspaces back."""
        for lane in self._lanes:
            if lane.pending is not None:
                lane.pending.result()
            if release:
                lane.ws = lane.raw_buf = None

    def _jitter(self, shape, device):
        return torch.rand(shape, device=device, dtype=torch.float32)

    # ---- train-mode forward (inb_renderer.py:78-103, inb_part_network_multiassign.py:162-165) ------
    def _pair_noise(self, like):
        return torch.rand_like(like)                       # compute_val_pair_around_range (:41)

    def _pair_noise_dense(self, rows, device):
        """Fused path: one uniform [0,1)^3 draw per dense (survivor slot, part) row — the reference draws rand_like of the
        selected rows only (:41); the selection is made on the device, so every candidate row gets a draw."""
        return torch.rand(rows, 3, device=device, dtype=torch.float32)

    def _render_train_fused(self, batch, jitter):
        """Train-mode forward as ONE differentiable node (autograd.TrainRenderFn = invr_train_fwd / invr_train_bwd): no host
        synchronisation, regulariser reductions on the device.  The dynamic-shape outputs of the reference contract (resd,
        tpts, tocc, oresd) are read back from the workspace only if somebody asks for them (LazyTrainRet)."""
        from . import autograd as ag
        cfg, net = self.cfg, self.net
        S = int(cfg.N_samples)
        ray_o, ray_d, near, far = batch['ray_o'][0], batch['ray_d'][0], batch['near'][0], batch['far'][0]
        n = ray_o.shape[0]
        ctx = net.prepare(batch)
        max_active = 0                                            # capacity = every ray-sample (a training patch is small)
        noise = self._pair_noise_dense(n * S * NUM_PARTS, ray_o.device) if cfg.use_pair_reg else None
        params = [p for p in net.parameters() if p.requires_grad]
        rgb, acc, raw, dist, terms, occ, weights, z, stats = ag.TrainRenderFn.apply(
            net, ctx, getattr(net, '_grad_arena', None), ray_o, ray_d, near, far, S, jitter, noise, max_active, *params)
        self.last_stats = stats
        self.last_train = {'weights': weights, 'z_vals': z, 'terms': terms}
        base = {'rgb_map': rgb[None], 'acc_map': acc[None], 'raw': raw[None], 'occ': occ[None, :, None]}
        if cfg.use_reg_distortion:
            base['reg_distortion_loss'] = dist[None]
        # inb_trainer.py:89-92 / :45-48 + crit.py:8-18 as differentiable scalars (sum / device-side count) — built when somebody reads
        # them: the trainer's fused objective (autograd.TrainLossFn) takes `terms` itself
        thunks = {'offset_loss': lambda: terms[ag.TERM_OFFSET_SUM] / terms[ag.TERM_OFFSET_ROWS].clamp(min=1.0)}
        lazy = ['resd', 'tpts', 'tocc']
        if cfg.use_pair_reg:
            thunks['pair_loss'] = lambda: terms[ag.TERM_PAIR_SUM] / terms[ag.TERM_PAIR_ROWS].clamp(min=1.0)
            lazy.append('oresd')
        ws, ws_gen = net._ws, net._ws_gen

        def materialise():
            ag.check_workspace(net, ws, ws_gen, "forward's resd / tpts / tocc / oresd read-back")
            with torch.no_grad():
                return self._train_extras(net, ctx, ws, stats, n, S, max_active, noise, (ray_o, ray_d, near, far), jitter)
        return ag.LazyTrainRet(base, lazy, materialise, thunks)

    def _train_extras(self, net, ctx, ws, stats_dev, n, S, max_active, noise_dense, rays, jitter):
        """resd / tpts / tocc (dense (Na*P, .) layouts in the reference's row order) and oresd from the pair lists of the
        last forward (host read-back of the counts: synchronises)."""
        cfg = self.cfg
        stats = stats_dev.cpu()
        v = _abi.ws_views(ws, n, S, max_active)
        dev = ws.device
        resd, tpts, tocc = dense_train_rows(v, stats, ctx, rays, S, jitter)
        out = {'resd': resd.reshape(1, -1, 3), 'tpts': tpts.reshape(1, -1, 3), 'tocc': tocc.reshape(1, -1, 1)}
        if cfg.use_pair_reg and noise_dense is not None:
            reg = ((tocc.reshape(-1) - 0.5).abs() < 0.02).nonzero(as_tuple=True)[0]
            if reg.numel():
                reg_tpts = tpts.reshape(-1, 3)[reg][None]
                neighbor = reg_tpts + (noise_dense[reg][None] - 0.5) * 0.01
                out['oresd'] = torch.cat([resd.reshape(-1, 3)[reg][None], net.resd(neighbor, ctx)], dim=1)
            else:
                out['oresd'] = torch.zeros(1, 0, 3, device=dev)
        return out

    def _render_train(self, batch, jitter):
        """Forward quantities of a training step: rgb_map/acc_map/raw/occ plus the train-only outputs
        resd, tpts, tocc (dense (Na*P, .) layouts in the reference's row order), oresd (pair
        regulariser) and reg_distortion_loss.  With gradients enabled: the fused HIP forward/backward node
        (_render_train_fused, cfg.train_fused, default) or — cfg.train_fused False — the op-by-op autograd graph of
        autograd.render_train on the pair lists (every dense output differentiable; the in-repo cross-check)."""
        cfg, net = self.cfg, self.net
        S = int(cfg.N_samples)
        ray_o, ray_d, near, far = batch['ray_o'][0], batch['ray_d'][0], batch['near'][0], batch['far'][0]
        n = ray_o.shape[0]
        if torch.is_grad_enabled() and cfg.get('train_fused', True):
            return self._render_train_fused(batch, jitter)
        ctx = net.prepare(batch)
        if torch.is_grad_enabled():       # geometry only: the differentiable part is recomputed on the pair lists
            out = net.geometry_pass(ctx, ray_o, ray_d, near, far, S, jitter=jitter)
        else:
            out = net.render_rays(ctx, ray_o, ray_d, near, far, S, jitter=jitter, want_raw=True, want_weights=True)
        stats = out['stats'].cpu()                          # host sync, as the reference's nonzero()s
        assert int(stats[6]) == 0, 'invr workspace overflow'
        Na = int(stats[0])
        ws, _, _, max_active = out['_ws']
        v = _abi.ws_views(ws, n, S, max_active)
        cap, dev = v['cap'], ray_o.device
        P = NUM_PARTS
        self.last_stats = out['stats']
        if torch.is_grad_enabled():
            # differentiable recomputation on the pair lists (autograd.py): HIP encoder / compositing
            # forward+backward kernels, torch for the tiny MLPs
            from . import autograd as ag
            return ag.render_train(net, batch, out, v, stats, n, S, self._pair_noise, ctx=ctx, rays=(ray_o, ray_d, near, far), jitter=jitter)
        resd, tpts, tocc = dense_train_rows(v, stats, ctx, (ray_o, ray_d, near, far), S, jitter)
        ret = {'rgb_map': out['rgb_map'][None], 'acc_map': out['acc_map'][None], 'raw': out['raw'][None],
               'occ': out['occ'][None, :, None], 'resd': resd.reshape(1, -1, 3), 'tpts': tpts.reshape(1, -1, 3),
               'tocc': tocc.reshape(1, -1, 1)}
        if cfg.use_pair_reg:                                                             # inb_renderer.py:78-94
            tflat = tocc.reshape(-1)
            reg = ((tflat - 0.5).abs() < 0.02).nonzero(as_tuple=True)[0]
            if reg.numel():
                reg_tpts = tpts.reshape(-1, 3)[reg][None]
                reg_resd = resd.reshape(-1, 3)[reg][None]
                neighbor = reg_tpts + (self._pair_noise(reg_tpts) - 0.5) * 0.01
                nei = net.resd(neighbor, ctx)
                ret['oresd'] = torch.cat([reg_resd, nei], dim=1)
            else:
                ret['oresd'] = torch.zeros(1, 0, 3, device=dev)
        if cfg.use_reg_distortion:                                                       # :96-103
            dl = torch.empty(n, device=dev)
            _abi.check(_abi.lib().invr_distortion_fwd(_abi.ptr(out['weights']), _abi.ptr(out['z_vals']), n, S,
                                                      _abi.ptr(dl), _abi.stream_ptr()))
            ret['reg_distortion_loss'] = dl[None]
        self.last_stats = out['stats']
        self.last_train = {'weights': out['weights'], 'z_vals': out['z_vals']}
        return ret
