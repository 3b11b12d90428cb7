"""Host-side mirror of the reference's network plug-in (lib/networks/bw_deform/
inb_part_network_multiassign.py:68-256, part_base_network.py:11-63, embedders/
part_base_embedder.py:13-104, embedders/freq_embedder.py:5-43, deformers/uv_deformer.py:12-21).

These are ``nn.Module`` containers only: they own the parameters under the reference's exact
attribute names (so ``state_dict()`` / ``load_state_dict()`` interchange ``.pth`` files with the
reference) and hand raw device pointers to libinvr.so.  All arithmetic of the path runs in the
HIP kernels; there is no PyTorch fallback.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _abi, params
from .config import cfg as global_cfg, PART_NAMES, NUM_PARTS, validate as validate_cfg


class Embedder(nn.Module):
    """Parameter container of one multi-resolution grid (part_base_embedder.py:13-104)."""

    def __init__(self, spec, pid=-1, partname='undefined'):
        super().__init__()
        self.spec, self.pid, self.partname = spec, pid, partname
        L, F, T = spec['L'], spec['F'], spec['T']
        ng = dict(requires_grad=False)
        self.bounds = nn.Parameter(torch.from_numpy(spec['bbox'].copy()), **ng)
        self.entries_size = nn.Parameter(torch.from_numpy(spec['size'].copy()), **ng)
        self.entries_num = nn.Parameter(torch.tensor(spec['res'], dtype=torch.int64), **ng)
        self.entries_min = nn.Parameter(torch.zeros(L, dtype=torch.int64), **ng)
        self.entries_cnt = nn.Parameter(torch.tensor(spec['cnt'], dtype=torch.int64), **ng)
        self.entries_sum = nn.Parameter(torch.tensor(spec['cnt'], dtype=torch.int64).cumsum(0), **ng)
        self.start_hash, self.n_levels, self.f, self.out_dim = spec['start_hash'], L, F, spec['out_dim']
        self.n_entries_per_level = T
        self.separate_dense = spec['separate_dense']
        self.use_batch_bounds = spec['use_batch_bounds']
        if self.separate_dense:
            data = torch.zeros((L, T, F))
            nn.init.kaiming_normal_(data)                    # one (L,T,F) init, then split (:70-76)
            self.dense = nn.Parameter(torch.cat([data[i, :spec['cnt'][i]] for i in range(self.start_hash)], 0))
            self.hash = nn.Parameter(data[self.start_hash:].clone())
        else:
            self.hash = nn.Parameter(torch.zeros((L, T, F)))
            nn.init.kaiming_normal_(self.hash)
        self.offsets = nn.Parameter(torch.from_numpy(params.CORNER_OFFSETS.copy()), **ng)

    def __getstate__(self):
        d = self.__dict__.copy()
        for k in ('_rs', '_rs_key', '_row_grad', 'row_grad_dirty'):
            d.pop(k, None)
        return d

    def maybe_adopt_batch_bounds(self, batch):
        # part_base_embedder.py:107-109: bounds are re-created from the batch at iter_step == 1
        if self.use_batch_bounds and 'iter_step' in batch and batch['iter_step'] == 1:
            self.bounds = nn.Parameter(batch['bounds'][0][self.pid].detach().clone(), requires_grad=False)

    def grid_struct(self, keep):
        return _abi.make_grid(self.spec, self.dense if self.separate_dense else None, self.hash, self.bounds, keep)

    def invalidate_row_sums(self):
        """Drop the cached row-sum table (rebuilt by the next eval render: one cheap kernel).  Call after writing the
        tables through a path the tensor version counters do not see (p.data.copy_(), raw-pointer kernels)."""
        self._rs_key = None
        self._rs = None

    def train(self, mode=True):
        # every train <-> eval transition rebuilds the derived table: in-place writes through .data (its own version
        # counter) or external kernels during training would otherwise leave eval rendering from stale row sums
        self.invalidate_row_sums()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_row_sums()
        return super()._load_from_state_dict(*args, **kwargs)

    def row_sums(self):
        """Inference-only derived table (invr_grid_row_sums): one float per table row = sum of its F
        features.  Cached; rebuilt when the tables were written to (tensor version counters) or moved, on every
        train()/eval() switch, on load_state_dict and after invalidate_row_sums()."""
        tabs = [self.hash] + ([self.dense] if self.separate_dense else [])
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in tabs)
        if getattr(self, '_rs_key', None) != key:
            keep = []
            g = self.grid_struct(keep)
            n = _abi.lib().invr_grid_row_sums_len(C.byref(g))
            rs = torch.empty(n, device=self.hash.device, dtype=torch.float32)
            _abi.check(_abi.lib().invr_grid_row_sums(C.byref(g), _abi.ptr(rs), _abi.stream_ptr()))
            self._rs, self._rs_key = rs, key
        return self._rs

    def row_grad(self):
        """Persistent compact table gradient of a sum-over-features grid (invr_train_bwd): one float per table row in
        invr_grid_row_sums order; zero-initialised, accumulated by the fused training backward, consumed and re-zeroed by
        FusedAdam (optim.py).  The dense gradient tensors are d(row)[f] = row_grad[row] for all F features."""
        rg = getattr(self, '_row_grad', None)
        if rg is None or rg.device != self.hash.device:
            keep = []
            g = self.grid_struct(keep)
            n = _abi.lib().invr_grid_row_sums_len(C.byref(g))
            rg = torch.zeros(n, device=self.hash.device, dtype=torch.float32)
            self._row_grad = rg
            self.row_grad_dirty = False
        return rg

    def expand_row_grad(self, accumulate=False):
        """Dense .grad tensors of the tables from the row-scalar gradient (invr_expand_row_grad)."""
        keep = []
        g = self.grid_struct(keep)
        gh = torch.empty_like(self.hash)
        gd = torch.empty_like(self.dense) if self.separate_dense else None
        _abi.check(_abi.lib().invr_expand_row_grad(C.byref(g), _abi.ptr(self.row_grad()), _abi.ptr(gd), _abi.ptr(gh), _abi.stream_ptr()))
        for p_, g_ in ((self.hash, gh),) + (((self.dense, gd),) if self.separate_dense else ()):
            p_.grad = g_ if (p_.grad is None or not accumulate) else p_.grad + g_
        return gd, gh

    def forward(self, xyz, batch=None):
        """HashEmbedder.forward (:106-174) through invr_grid_encode_fwd."""
        if batch is not None:
            self.maybe_adopt_batch_bounds(batch)
        keep = []
        g = self.grid_struct(keep)
        x = xyz.detach().to(torch.float32).contiguous()
        out = torch.empty(x.shape[0], self.out_dim, device=x.device, dtype=torch.float32)
        _abi.check(_abi.lib().invr_grid_encode_fwd(C.byref(g), _abi.ptr(x), x.shape[0], _abi.ptr(out), _abi.stream_ptr()))
        return out


class _PosEnc(nn.Module):
    def __init__(self, multires):
        super().__init__()
        fb = 2. ** torch.linspace(0., multires - 1, steps=multires)
        self.freq_bands = nn.Parameter(fb[..., None, None].expand(multires, 2, 1).clone(), requires_grad=False)
        self.multires = multires


class DirEmbedder(nn.Module):
    """freq_embedder.Embedder container (:38-43); evaluated inside the part MLP kernel."""

    def __init__(self, res, input_dims=3):
        super().__init__()
        self.embedder = _PosEnc(res)
        self.out_dim = input_dims + input_dims * 2 * res


class MLP(nn.Module):
    """part_base_network.MLP container (:11-24)."""

    def __init__(self, indim=16, outdim=3, d_hidden=64, n_layers=2):
        super().__init__()
        self.indim, self.outdim = indim, outdim
        self.linears = nn.ModuleList([nn.Linear(indim, d_hidden)] + [nn.Linear(d_hidden, d_hidden) for _ in range(n_layers - 1)]
                                     + [nn.Linear(d_hidden, outdim)])
        self.actvn = nn.Softplus()

    def forward(self, x):
        """torch path (training autograd only; inference runs in k_part_mlp)."""
        for l in self.linears[:-1]:
            x = self.actvn(l(x))
        return self.linears[-1](x)


ColorNetwork = MLP


class PartNetwork(nn.Module):
    """part_base_network.Network container (:31-42)."""

    def __init__(self, partname, pid, cfg=None):
        super().__init__()
        cfg = cfg or global_cfg
        self.pid, self.partname = pid, partname
        self.embedder = Embedder(params.part_grid_spec(cfg, partname), pid, partname)
        self.embedder_dir = DirEmbedder(**cfg.viewdir_embedder.kwargs)
        occ_dims, rgb_dims = params.mlp_dims(cfg, partname)
        self.occ = MLP(occ_dims[0], occ_dims[-1], cfg.network.occ['d_hidden'], cfg.network.occ['n_layers'])
        self.rgb_latent = nn.Parameter(torch.zeros(cfg.num_latent_code, cfg.latent_code_dim))
        nn.init.kaiming_normal_(self.rgb_latent)
        ck = cfg.partnet[partname].color_network.kwargs
        self.rgb = ColorNetwork(rgb_dims[0], 3, ck['d_hidden'], ck['n_layers'])


class Deformer(nn.Module):
    """uv_deformer.Deformer container (:12-21)."""

    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or global_cfg
        self.embedder = Embedder(params.deformer_grid_spec(cfg))
        self.mlp = nn.Sequential(nn.Linear(self.embedder.out_dim, 32), nn.Softplus(), nn.Linear(32, 32),
                                 nn.Softplus(), nn.Linear(32, 3))


class TPoseHuman(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        self.part_networks = nn.ModuleList([PartNetwork(p, i, cfg) for i, p in enumerate(PART_NAMES)])


class RenderContext:
    def __init__(self, model, scene, keep):
        self.model, self.scene, self.keep = model, scene, keep


class Network(nn.Module):
    """Drop-in for inb_part_network_multiassign.Network (:68-168)."""

    def __init__(self, init_network=True, cfg=None):
        super().__init__()
        self.cfg = cfg or global_cfg
        validate_cfg(self.cfg)                  # raises on reference switches this build does not implement
        self.tpose_deformer = Deformer(self.cfg)
        self.tpose_human = TPoseHuman(self.cfg)
        self._ws = None

    _TRANSIENT = ('_model_key', '_model_base', '_model_keep', '_ws', '_ws_gen', '_grad_arena')

    def __getstate__(self):
        # derived ctypes views / scratch buffers are not part of the module's state (copy.deepcopy, pickling)
        d = self.__dict__.copy()
        for k in self._TRANSIENT:
            d.pop(k, None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._ws = None

    # -- C-ABI glue ---------------------------------------------------------------------------
    def model_struct(self, keep):
        # the ctypes view of the 186 parameter tensors is rebuilt only when a storage moved (load_state_dict, .to(),
        # bounds adoption): building it costs ~1 ms of host time, a third of a training iteration's budget
        params_now = list(self.named_parameters())
        key = tuple(p.data_ptr() for _, p in params_now)
        if getattr(self, '_model_key', None) != key:
            self._model_keep = []
            self._model_base = _abi.make_model(dict(params_now), self.cfg, self._model_keep)
            self._model_key = key
        keep.extend(self._model_keep)
        m = _abi.InvrModel.from_buffer_copy(self._model_base)
        if not self.training and self.cfg.get('eval_row_sums', True):
            # eval: the per-part grids are read through their row-sum tables (16x fewer table bytes)
            for i, pn in enumerate(self.tpose_human.part_networks):
                e = pn.embedder
                if e.spec['sum'] and e.spec['sum_over_features'] and e.f % 4 == 0:
                    rs = e.row_sums()
                    keep.append(rs)
                    m.part[i].grid.row_sums = rs.data_ptr()
        return m

    def workspace(self, nbytes, device):
        """The shared scratch buffer of this network's library calls.  Every hand-out bumps `_ws_gen`: a caller that reads the
        buffer again later (the training backward, the lazily materialised train-mode tensors) remembers the generation of its
        own call and refuses to run on a buffer some other call has overwritten since."""
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._ws_gen = getattr(self, '_ws_gen', 0) + 1
        return self._ws

    def adopt_batch_bounds(self, batch):
        for pn in self.tpose_human.part_networks:
            pn.embedder.maybe_adopt_batch_bounds(batch)

    def prepare(self, batch):
        """Build the C-ABI views (InvrModel / InvrScene) of the parameters and of one batch.
        The result can be reused for any number of render_rays calls while neither changes."""
        self.adopt_batch_bounds(batch)
        keep = []
        return RenderContext(self.model_struct(keep), _abi.make_scene(batch, self.cfg, keep), keep)

    def forward(self, wpts, viewdir, dists, batch):
        """Network.forward (inb_part_network_multiassign.py:126-168), eval outputs: world points
        (N,3), view directions (N,3) -> {'raw': (1,N,4), 'occ': (1,N,1)}.  `dists` is unused, as in the
        reference.  Training gradients flow through Renderer.render (autograd.py), not through here."""
        L = _abi.lib()
        ctx = batch if isinstance(batch, RenderContext) else self.prepare(batch)
        x = wpts.detach().to(torch.float32).contiguous()
        d = viewdir.detach().to(torch.float32).contiguous()
        n = x.shape[0]
        raw = torch.empty(n, 4, device=x.device)
        occ = torch.empty(n, device=x.device)
        stats = torch.zeros(_abi.STATS_LEN, dtype=torch.int32, device=x.device)
        nbytes = L.invr_field_workspace_bytes(n, 0)
        ws = self.workspace(nbytes, x.device)
        _abi.check(L.invr_field_fwd(C.byref(ctx.scene), C.byref(ctx.model), _abi.ptr(x), _abi.ptr(d), n, _abi.ptr(raw),
                                    _abi.ptr(occ), _abi.ptr(stats, torch.int32), C.c_void_p(ws.data_ptr()), nbytes, 0,
                                    _abi.stream_ptr()))
        return {'raw': raw[None], 'occ': occ[None, :, None]}

    def resd(self, tpts, batch):
        """Network.resd (inb_part_network_multiassign.py:122-124): deformer residual of canonical
        points (B,N,3) -> (B,N,3), through invr_deform_fwd."""
        B, N, D = tpts.shape
        ctx = batch if isinstance(batch, RenderContext) else self.prepare(batch)
        x = tpts.detach().reshape(-1, 3).to(torch.float32).contiguous()
        out = torch.empty_like(x)
        _abi.check(_abi.lib().invr_deform_fwd(C.byref(ctx.scene), C.byref(ctx.model), _abi.ptr(x), x.shape[0],
                                              _abi.ptr(out), _abi.stream_ptr()))
        return out.view(B, N, D)

    def geometry_pass(self, batch, ray_o, ray_d, near, far, n_samples, jitter=None, max_active=0):
        """invr_geometry_fwd: the no-grad front half (pair lists in the workspace) for the training forward."""
        L = _abi.lib()
        dev = ray_o.device
        ctx = batch if isinstance(batch, RenderContext) else self.prepare(batch)
        f = lambda t: t.detach().to(torch.float32).contiguous()
        ray_o, ray_d, near, far = f(ray_o), f(ray_d), f(near), f(far)
        n, S = ray_o.shape[0], int(n_samples)
        out = {'z_vals': torch.empty(n, S, device=dev), 'stats': torch.zeros(_abi.STATS_LEN, dtype=torch.int32, device=dev)}
        if jitter is not None:
            jitter = f(jitter)
        nbytes = L.invr_workspace_bytes(n, S, max_active)
        ws = self.workspace(nbytes, dev)
        _abi.check(L.invr_geometry_fwd(C.byref(ctx.scene), C.byref(ctx.model), _abi.ptr(ray_o), _abi.ptr(ray_d), _abi.ptr(near),
                                       _abi.ptr(far), _abi.ptr(jitter), n, S, _abi.ptr(out['z_vals']),
                                       _abi.ptr(out['stats'], torch.int32), C.c_void_p(ws.data_ptr()), nbytes, max_active,
                                       _abi.stream_ptr()))
        out['_keep'] = ctx.keep
        out['_ws'] = (ws, n, S, max_active)
        return out

    def render_rays(self, batch, ray_o, ray_d, near, far, n_samples, jitter=None, want_raw=True,
                    want_weights=False, max_active=0, stream=None, raw_out=None):
        """One invr_render_fwd call over a ray list (n,3)/(n,).  `batch` is the collated batch dict
        or a RenderContext from prepare().  Returns a dict of device tensors.
        stream: launch on THAT torch stream instead of the current one while every tensor (outputs, workspace) still comes from the
        current stream's allocator pool (Renderer.in_flight lanes: K streams must not mean K private pools of 4 GB workspaces and
        0.5 GB raw tensors); the caller orders `stream` behind the current stream before the call, the tensors are recorded on it.
        raw_out: a flat float32 device buffer of at least n * S * 4 elements to hold `raw` (a lane's buffer nobody references any more)."""
        L = _abi.lib()
        dev = ray_o.device
        ctx = batch if isinstance(batch, RenderContext) else self.prepare(batch)
        model, scene, keep = ctx.model, ctx.scene, ctx.keep
        f = lambda t: t.detach().to(torch.float32).contiguous()
        ray_o, ray_d, near, far = f(ray_o), f(ray_d), f(near), f(far)
        n = ray_o.shape[0]
        S = int(n_samples)
        # (the library writes every entry of stats unless there is nothing to render)
        out = {'rgb_map': torch.empty(n, 3, device=dev), 'acc_map': torch.empty(n, device=dev),
               'stats': (torch.empty if n else torch.zeros)(_abi.STATS_LEN, dtype=torch.int32, device=dev)}
        if want_raw:
            out['raw'] = torch.empty(n * S, 4, device=dev) if raw_out is None else raw_out[:n * S * 4].view(n * S, 4)
        if want_weights:
            out['weights'] = torch.empty(n, S, device=dev)
            out['z_vals'] = torch.empty(n, S, device=dev)
        if jitter is not None:
            jitter = f(jitter)
        nbytes = L.invr_workspace_bytes(n, S, max_active)
        ws = self.workspace(nbytes, dev)
        _abi.check(L.invr_render_fwd(
            C.byref(scene), C.byref(model), _abi.ptr(ray_o), _abi.ptr(ray_d), _abi.ptr(near), _abi.ptr(far),
            _abi.ptr(jitter), n, S, _abi.ptr(out['rgb_map']), _abi.ptr(out['acc_map']),
            _abi.ptr(out.get('raw')), _abi.ptr(None), _abi.ptr(out.get('weights')),
            _abi.ptr(out.get('z_vals')), _abi.ptr(out['stats'], torch.int32),
            C.c_void_p(ws.data_ptr()), nbytes, max_active, _abi.stream_ptr() if stream is None else C.c_void_p(stream.cuda_stream)))
        if stream is not None:
            for t in (ws, ray_o, ray_d, near, far, jitter) + tuple(v for v in out.values() if torch.is_tensor(v)):
                if t is not None:
                    t.record_stream(stream)
        if want_raw:
            out['occ'] = out['raw'][:, 3]          # occ IS raw's fourth channel (inb_part_network_multiassign.py:229-256 returns both from the
                                                   # same rows): a view, not a second 4 B / sample write of the compositing kernel
        out['_keep'] = keep
        out['_ws'] = (ws, n, S, max_active)
        return out
