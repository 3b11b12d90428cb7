"""Frames in flight: K frames of a sequence rendered by ONE hipGraph replay.

Why.  A ray shard of a frame is a chain of ~14 dependent kernels, each with a ramp, a latency-bound tail and (the KNN) whole-CU
LDS footprints; on a 1/8 shard half of the chip idles at any moment (profiles/r4_front_chain.md).  Frames of a sequence are
independent, so their chains can fill each other's gaps — but only inside one graph: two graph launches on two streams do not
overlap on this runtime, parallel BRANCHES of one graph do.  `FrameSet` captures K render calls as K branches (each on its own
side stream, with its own workspace and outputs), joins them, and — with more than one rank — appends ONE all-gather of the K
frames' [r, g, b, acc] tiles (RCCL, captured in the same graph) and the K index_selects that put the rows in ray order.
`replay()` is one launch per K frames: no per-frame host work, no per-frame collective.
Measured on one MI355X (rank 0's shard of a W-way split, ms per frame): W = 1: 2.38 -> 2.08 (K = 4) -> 2.02 (K = 10), W = 8: 0.50 -> 0.33.

The reference has no equivalent: its renderer walks one frame in 4096-ray chunks (inb_renderer.py:217-237) and its only
parallelism is DDP training."""
import torch
import torch.distributed as dist

from .dist import DEFAULT_TILE, FORCE_COLLECTIVES, gather_plan


def _packet_capture_off():
    import sys
    return bool(getattr(sys.modules.get('invr'), 'PACKET_CAPTURE_OFF', False))


def exchange_plan(n_rays, world, tile=DEFAULT_TILE, device='cpu'):
    """Layout of ONE all-gather for K frames: every rank sends `rows` = sum_k mx_k rows (frame k at row offset off[k], padded to
    the largest shard mx_k of that frame); ray i of frame k then sits at row src[k][i] of the gathered (world * rows, 4) buffer."""
    off, src, mxs, rows = [], [], [], 0
    for n in n_rays:
        mx, s = gather_plan(int(n), world, tile, device)
        off.append(rows)
        mxs.append(mx)
        rows += mx
    for k, n in enumerate(n_rays):
        mx, s = gather_plan(int(n), world, tile, device)
        src.append((s // mx) * rows + off[k] + (s % mx))
    return {'rows': rows, 'off': off, 'mx': mxs, 'src': src}


class FrameSet:
    """K frames per replay.  `render_fns[k]()` renders this rank's rays of frame k on the CURRENT stream and returns either the
    (n_local_k, 4) [r, g, b, acc] rows or a dict with 'rgb_map' (n,3) and 'acc_map' (n,); `n_rays[k]` = rays of the whole frame k.
    After `replay()`: `local[k]` = what render_fns[k] returned, `full[k]` = the (n_rays[k], 4) map of the whole frame on every rank
    (`local` rows for a group of one).  capture=False runs the same steps eagerly (CPU / gloo tests, debugging); capture_exchange=False
    captures the renders only and issues the all-gather + index_selects from the host behind every replay (one per K frames)."""

    def __init__(self, render_fns, n_rays, rank=0, world=1, device='cuda', tile=DEFAULT_TILE, group=None, capture=True, capture_exchange=True,
                 streams=False):
        self.fns, self.n_rays, self.rank, self.world = list(render_fns), [int(n) for n in n_rays], rank, world
        self.device, self.tile, self.group, self.capture = torch.device(device), tile, group, capture
        self.K = len(self.fns)
        self.use_streams = bool(streams) and not capture and self.device.type == 'cuda'
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.K)] if self.use_streams else None
        self.exchange = world > 1 or (FORCE_COLLECTIVES() and dist.is_initialized())
        self.plan = exchange_plan(self.n_rays, world, tile, self.device) if self.exchange else None
        self.graph, self.exchange_captured, self.want_exchange_captured = None, False, capture_exchange
        self.local, self.full = [None] * self.K, [None] * self.K
        if self.exchange:
            self.send = torch.zeros(self.plan['rows'], 4, device=self.device)
            self.recv = torch.empty(world * self.plan['rows'], 4, device=self.device)
        if capture:
            self._capture()

    def own_rows_match(self):
        """Sanity check of an exchange: the rows of every full map that THIS rank rendered equal its local rows (synchronises)."""
        from .dist import tile_indices
        for k in range(self.K):
            if self.full[k] is None or self.local[k] is None:
                return False
            idx = tile_indices(self.n_rays[k], self.rank, self.world, self.tile, device=self.full[k].device)
            if not torch.equal(self.full[k][idx], self._rgba(self.local[k])):
                return False
        return True

    @staticmethod
    def _rgba(out):
        return out if torch.is_tensor(out) else torch.cat([out['rgb_map'], out['acc_map'][:, None]], 1)

    def _render(self, k):
        out = self.fns[k]()
        self.local[k] = out
        rgba = self._rgba(out)
        if self.exchange:
            o = self.plan['off'][k]
            self.send[o:o + rgba.shape[0]].copy_(rgba)
        else:
            self.full[k] = rgba

    def _exchange(self):
        if not self.exchange:
            return
        backend = dist.get_backend(self.group)
        if backend == 'gloo' and self.send.is_cuda:              # gloo (tests) gathers through host memory
            host = torch.empty(self.recv.shape, dtype=self.recv.dtype)
            dist.all_gather_into_tensor(host, self.send.cpu(), group=self.group)
            self.recv.copy_(host)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        for k in range(self.K):
            self.full[k] = self.recv.index_select(0, self.plan['src'][k])

    def _capture(self):
        assert self.device.type == 'cuda', 'graph capture needs a GPU (capture=False runs eagerly)'
        # the exchange is part of the graph on RCCL; any other backend (gloo in 1-GPU debugging runs) exchanges eagerly behind the replay
        self.exchange_captured = self.exchange and self.want_exchange_captured and dist.get_backend(self.group) == 'nccl'
        # warm-up outside the capture (workspace allocation, lazy module state, RCCL communicator set-up)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for k in range(self.K):
                self._render(k)
            self._exchange()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.K)]          # capture streams of the K branches
        # thread_local capture mode: the RCCL watchdog thread of a multi-rank run may query events while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            cur = torch.cuda.current_stream(self.device)
            # every frame on a side stream of its own (a branch left on the capture stream itself does not overlap the others)
            for k in range(self.K):
                self.streams[k].wait_stream(cur)
                with torch.cuda.stream(self.streams[k]):
                    self._render(k)
            for k in range(self.K):
                cur.wait_stream(self.streams[k])
            if self.exchange and self.K == 1 and not _packet_capture_off():
                # one frame + a collective = a graph without parallel branches, which this runtime replays from pre-built packets —
                # the path that faults beside RCCL (invr/__init__.py) unless it was switched off before HIP initialised: give the
                # graph a second, empty branch instead
                self._side = getattr(self, '_side', None) or torch.zeros(64, device=self.device)
                extra = torch.cuda.Stream(self.device)
                self.streams.append(extra)
                extra.wait_stream(cur)
                with torch.cuda.stream(extra):
                    self._side.add_(0.0)
                cur.wait_stream(extra)
            if self.exchange_captured:
                self._exchange()
        torch.cuda.synchronize(self.device)

    def replay(self):
        if self.graph is not None:
            self.graph.replay()
            if self.exchange and not self.exchange_captured:
                self._exchange()
            return
        if self.use_streams:
            # no graph: frame k's chain is launched eagerly on stream k (its workspace and outputs are only ever touched there, so
            # replay r + 1 of frame k queues behind replay r of frame k and nothing else); with an exchange the renders first wait
            # for the previous replay's exchange to have read the send buffer, and the exchange waits for the K renders
            cur = torch.cuda.current_stream(self.device)
            for k in range(self.K):
                self.streams[k].wait_stream(cur)          # (inputs made on the caller's stream; with an exchange: the previous replay's gather)
                with torch.cuda.stream(self.streams[k]):
                    self._render(k)
            # the caller's stream is ordered behind the K chains in every case: what replay() leaves in `local` / `full`
            # may be read on the current stream right away, as after a graph replay or the eager loop (K event waits, no host wait)
            for k in range(self.K):
                cur.wait_stream(self.streams[k])
            if self.exchange:
                self._exchange()
            return
        for k in range(self.K):
            self._render(k)
        self._exchange()

    def synchronize(self):
        """host-side join of everything replay() has enqueued (the K streams of the no-graph mode included)"""
        torch.cuda.synchronize(self.device) if self.device.type == 'cuda' else None


def shard_render_fns(net, batches, n_samples, rank, world, tile=DEFAULT_TILE, want_raw=True, cap_margin=1.3):
    """render_fns / n_rays for FrameSet from collated batches on the device: frame k's rays are dealt tile-cyclically, rank's shard is
    rendered by net.render_rays with a workspace of its own, sized from the survivor count of a first render (cap_margin x, as
    Renderer does frame to frame; an overflow shows in stats[6] — check_overflow)."""
    from .dist import tile_indices
    fns, n_rays, keep = [], [], []
    for b in batches:
        dev = b['ray_o'].device
        n = int(b['ray_o'].shape[1])
        idx = tile_indices(n, rank, world, tile, device=dev)
        a = tuple(b[k][0][idx].contiguous() for k in ('ray_o', 'ray_d', 'near', 'far'))
        ctx = net.prepare(b)
        net._ws = None
        st = net.render_rays(ctx, a[0], a[1], a[2], a[3], n_samples, want_raw=False)['stats'].cpu()
        cap = int(min(a[0].shape[0] * n_samples, max(65536, -(-int(float(st[0]) * cap_margin) // 65536) * 65536)))
        net._ws = None
        net.render_rays(ctx, a[0], a[1], a[2], a[3], n_samples, want_raw=want_raw, max_active=cap)     # allocates this frame's workspace
        ws = net._ws
        net._ws = None

        def fn(ctx=ctx, a=a, ws=ws, cap=cap):
            net._ws = ws
            out = net.render_rays(ctx, a[0], a[1], a[2], a[3], n_samples, want_raw=want_raw, max_active=cap)
            net._ws = None
            return out
        fns.append(fn)
        n_rays.append(n)
        keep.append((ctx, a, ws))
    return fns, n_rays, keep


def check_overflow(frame_set):
    """True if no frame of the last replay ran over its workspace capacity (reads the statistics blocks: synchronises)."""
    return all(int(o['stats'][6]) == 0 for o in frame_set.local if isinstance(o, dict) and 'stats' in o)
