"""Row f1: the optimiser of the reference's training loop as one HIP launch.

`FusedAdam` is torch.optim.Adam (lib/train/optimizer.py:13-31 builds it with one parameter group per tensor,
`eps=cfg.train.eps`) with the update of every tensor done by `invr_adam_step`; hyper-parameters, per-group `lr`
(the reference's schedulers write `group['lr']`), `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq`) and the
skip-tensors-without-gradient rule (`p.grad is None`) are torch's, so optimiser checkpoints interchange with the reference's.
On the fused path every tensor takes every step — also the tensors of a body part without a flagged pair in the batch: the
reference calls all five part networks even on zero points (inb_part_network_multiassign.py:223-229), their gradients are
zeros, not None, and its Adam advances `step`, decays the moments and moves the parameters by momentum.

`attach(net)` switches the network's training path to a persistent gradient arena (autograd.GradArena): the fused
backward (invr_train_bwd) accumulates into fixed addresses — the five part grids as compact ROW-SCALAR gradients, 68 MB
instead of 1.09 GB (the gradient of a sum-over-features table is one scalar per row) — so the device table of this
optimiser is uploaded once and an iteration is {invr_adam_advance, invr_adam_step} with no host-built data, no 1.09 GB
zero-fill and 24.25 instead of 28 bytes of HBM traffic per parameter.  Adam itself stays dense: rows without a gradient
this step still move through their first moment, as in the reference.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _abi


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plan_key = None
        self._pending = 0             # steps taken on the device since the host-side `step` tensors were last written
        self._plan_params = []
        self.arena = None

    # ---- fused-path gradient arena ------------------------------------------------------------------------------
    def attach(self, net):
        """Give `net` (invr.network.Network) a persistent gradient arena consumed by this optimiser."""
        from .autograd import GradArena
        self.arena = GradArena(net)
        net._grad_arena = self.arena
        self._row_grad_of = {}
        for e in self.arena.embedders:
            rows_dense = e.dense.shape[0] if e.separate_dense else 0
            rg = e.row_grad()
            if e.separate_dense:
                self._row_grad_of[id(e.dense)] = (e, rg[:rows_dense])
            self._row_grad_of[id(e.hash)] = (e, rg[rows_dense:])
        return self

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none)
        if self.arena is not None:
            self.arena.zero()

    def _flush_steps(self):
        """Write the device-side step counts back into the host `step` tensors (state_dict layout of torch.optim.Adam)."""
        if self._pending:
            tab = (_abi.InvrAdamTensor * len(self._plan_params)).from_buffer_copy(bytes(self._table.cpu().numpy()))
            for p, e in zip(self._plan_params, tab):
                self.state[p]['step'] = torch.tensor(float(e.step))
            self._pending = 0

    def state_dict(self):
        self._flush_steps()
        return super().state_dict()

    def __getstate__(self):
        # copy.deepcopy / pickle read `state` directly: make the host-side step counts current first.  (Reading
        # optimizer.state[p]['step'] by hand between steps sees the value of the last flush; a scheduler that changes a learning
        # rate EVERY iteration rebuilds the device table each time — one small device-to-host read per step.)
        self._flush_steps()
        return super().__getstate__()

    def load_state_dict(self, sd):
        self._flush_steps()
        super().load_state_dict(sd)
        self._plan_key = None

    # ---- one step -----------------------------------------------------------------------------------------------
    def _entries(self):
        """[(param, group, grad tensor, grad_shift)] of the tensors that have a gradient this step."""
        out = []
        betas = eps = None
        for group in self.param_groups:
            if betas is None:
                betas, eps = group['betas'], group['eps']
            assert (betas, eps) == (group['betas'], group['eps']), 'FusedAdam: betas / eps must be the same in all groups'
            for p in group['params']:
                g, shift = p.grad, 0
                if g is None and self.arena is not None and id(p) in self._row_grad_of:
                    e, rg = self._row_grad_of[id(p)]
                    if e.row_grad_dirty:                                  # the fused backward wrote row-scalar gradients
                        g, shift = rg, int(round(math.log2(e.f)))
                        assert (1 << shift) == e.f
                if g is None:
                    continue
                _abi.ptr(p)                                               # (the binding's checks: device tensor, float32, contiguous)
                assert not g.is_sparse
                out.append((p, group, g if g.is_contiguous() else g.contiguous(), shift))
        return out, betas, eps

    def _check_replicas(self):
        """With the arena the table gradients never pass through p.grad: under data parallelism somebody has to average the arena
        itself (dist_train.GradReducer).  A plain DistributedDataParallel wrapper would silently train every rank on its own
        table gradients."""
        if self.arena is None or getattr(self.arena, 'reducer', None) is not None or getattr(self, '_replicas_checked', False):
            return
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError('FusedAdam with a gradient arena in a process group of %d ranks but no invr.dist_train.GradReducer: the '
                               'row-scalar table gradients bypass p.grad, so DistributedDataParallel does not average them.  Use '
                               'dist_train (driver / bench --train --gpus N) or build the optimizer without attach().' % dist.get_world_size())
        self._replicas_checked = True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._check_replicas()
        entries, betas, eps = self._entries()
        if not entries:
            return loss
        key = tuple((p.data_ptr(), g.data_ptr(), p.numel(), group['lr'], group['weight_decay'], sh) for p, group, g, sh in entries)
        L = _abi.lib()
        if key != self._plan_key:
            # (re)build the device table: pointers, sizes, learning rates and the step counts so far.  With a gradient arena
            # this happens once (and when a scheduler changes a learning rate); with autograd-allocated gradients whenever an
            # address changes.
            self._flush_steps()
            tab = (_abi.InvrAdamTensor * len(entries))()
            E = L.invr_adam_chunk_elems()
            ct, ci = [], []
            for t, (e, (p, group, g, sh)) in enumerate(zip(tab, entries)):
                st = self.state[p]
                if not st:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                e.param, e.grad, e.exp_avg, e.exp_avg_sq = p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
                e.numel, e.lr, e.weight_decay = p.numel(), group['lr'], group['weight_decay']
                e.grad_shift, e.step = sh, int(st['step'])
                e.active = None
                n = (p.numel() + E - 1) // E
                ct.append(np.full(n, t, np.int32))
                ci.append(np.arange(n, dtype=np.int32))
            dev = entries[0][0].device
            self._table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
            self._chunk_tensor = torch.from_numpy(np.concatenate(ct)).to(dev)
            self._chunk_index = torch.from_numpy(np.concatenate(ci)).to(dev)
            self._plan_key = key
            self._plan_params = [p for p, _, _, _ in entries]
            self._plan_grads = [g for _, _, g, _ in entries]          # keep the gradient tensors of the table alive
        else:
            self._plan_grads = [g for _, _, g, _ in entries]
        _abi.check(L.invr_adam_advance(C.c_void_p(self._table.data_ptr()), len(entries), betas[0], betas[1], _abi.stream_ptr()))
        _abi.check(L.invr_adam_step(C.c_void_p(self._table.data_ptr()), _abi.ptr(self._chunk_tensor, torch.int32),
                                    _abi.ptr(self._chunk_index, torch.int32), self._chunk_tensor.numel(), betas[0], betas[1], eps,
                                    _abi.stream_ptr()))
        self._pending += 1
        # the kernel wrote through raw pointers: tell autograd / version-keyed caches (Embedder.row_sums)
        torch.autograd.graph.increment_version(self._plan_params)
        return loss


# Adam variants the fused step does not implement: such an optimizer is left alone (decoupled_weight_decay: torch >= 2.7 accepts it on
# plain torch.optim.Adam = AdamW semantics; the fused kernel applies L2 weight decay)
_UNSUPPORTED_ADAM_FLAGS = ('amsgrad', 'maximize', 'capturable', 'differentiable', 'decoupled_weight_decay')


def fuse(optimizer, net=None):
    """One line for a host that builds its optimiser the reference's way (lib/train/optimizer.py:13-31 -> torch.optim.Adam, one
    group per tensor):  `optimizer = invr.optim.fuse(optimizer, network)`  after make_optimizer.  Returns a FusedAdam over the SAME
    parameter groups (per-group lr / weight_decay, betas, eps) that continues from the optimiser's state (step counts, moments:
    state_dict layouts are identical), attached to `net`'s fused training path when it has one.  Anything that is not a plain
    torch.optim.Adam (amsgrad, maximize, capturable, a different class) is returned unchanged."""
    if isinstance(optimizer, FusedAdam) or type(optimizer) is not torch.optim.Adam:
        return optimizer
    groups = optimizer.param_groups
    if any(g.get(k) for g in groups for k in _UNSUPPORTED_ADAM_FLAGS):
        return optimizer
    if len({(tuple(g['betas']), g['eps']) for g in groups}) != 1:
        return optimizer
    try:                                        # (the binding's own checks: device tensors, float32, contiguous)
        for g in groups:
            for p in g['params']:
                _abi.ptr(p)
    except AssertionError:
        return optimizer
    d = optimizer.defaults
    new = FusedAdam([{'params': list(g['params']), 'lr': g['lr'], 'weight_decay': g['weight_decay'], 'betas': g['betas'], 'eps': g['eps']}
                     for g in groups], d['lr'], betas=d['betas'], eps=d['eps'], weight_decay=d['weight_decay'])
    # the SAME group dicts and the SAME state mapping as the optimizer that is replaced: a scheduler (or any wrapper) constructed on
    # the old object before fuse() keeps driving this one — it writes group['lr'] into dicts both share — and the moments /
    # step counts continue in place
    new.param_groups = optimizer.param_groups
    new.state = optimizer.state
    if net is not None and hasattr(net, 'tpose_human') and getattr(net, 'cfg', {}).get('train_fused', True):
        new.attach(net)
    return new


# ---- drop-in: the reference's own optimizer, unchanged, at the fused step's speed ---------------------------------------------
# train_net.py builds torch.optim.Adam over net.named_parameters() (lib/train/optimizer.py:15-31), hands it to its schedulers and to
# Trainer.train (trainer.py:139-149: zero_grad -> backward -> scaler.step(optimizer)).  With that optimizer an iteration is 9.3 ms —
# 1.09 GB of dense table gradients and torch's foreach Adam over 286 M parameters — against 3.7 ms with FusedAdam + the gradient
# arena.  `adopt_on_first_step(net)` (called by invr.trainer.NetworkWrapper) closes the gap without an edit of the host: a global
# optimizer-step pre-hook recognises, at its FIRST step, a plain torch.optim.Adam whose parameters are exactly this network's, and
# from then on that optimizer OBJECT (the one the host's schedulers and checkpoints hold) steps through invr_adam_step:
#   * an inner FusedAdam SHARES the host optimizer's `param_groups` list and `state` dict — a scheduler's group['lr'] writes are
#     seen, the moments live where optimizer.state_dict() / save_model (net_utils.py:461-479) expect them, same layout;
#   * the host object's step / zero_grad / state_dict / load_state_dict are bound to the inner optimizer's;
#   * the gradient arena is attached (row-scalar table gradients: no 1.09 GB dense gradient) unless a process group exists — a
#     DistributedDataParallel wrapper (trainer.py:21-26) reduces p.grad and expects every parameter to receive one, which the arena
#     bypasses — in which case the dense gradients stay and only the update is fused.
# The first step itself is taken by the fused kernel from the dense gradients of the first backward; torch's own step, which runs
# right after the hook, then finds no gradients and does nothing.  INVR_NO_OPTIM_HOOK=1 or cfg.fused_optimizer_hook False: off.
import os as _os
import types as _types
import weakref as _weakref

_ADOPT_NETS = _weakref.WeakSet()
_HOOK = [None]


def _adam_matches(opt, net):
    if type(opt) is not torch.optim.Adam or getattr(opt, '_invr_inner', None) is not None:
        return False
    groups = opt.param_groups
    if any(g.get(k) for g in groups for k in _UNSUPPORTED_ADAM_FLAGS):
        return False
    if len({(tuple(g['betas']), g['eps']) for g in groups}) != 1:
        return False
    mine = {id(p) for p in net.parameters() if p.requires_grad}
    theirs = [p for g in groups for p in g['params']]
    if {id(p) for p in theirs} != mine:
        return False
    return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in theirs)


def _adoption_supported():
    """The adoption rebinds step / zero_grad / state_dict / load_state_dict of the host's optimizer OBJECT and sets two marks torch's
    lr_scheduler looks for (`step._wrapped_by_lr_sched`, `opt._opt_called`): torch-private names, checked here against the running
    torch instead of assumed.  False -> the host's Adam is left alone (torch's own step: correct, slower) with ONE loud warning."""
    import inspect
    import warnings
    ok, why = True, ''
    try:
        from torch.optim import lr_scheduler as _ls
        src = inspect.getsource(_ls)
        for mark in ('_wrapped_by_lr_sched', '_opt_called'):
            if mark not in src:
                ok, why = False, 'torch.optim.lr_scheduler no longer uses %r' % mark
        sig = inspect.signature(torch.optim.Adam.zero_grad)
        if 'set_to_none' not in sig.parameters:
            ok, why = False, 'Optimizer.zero_grad lost set_to_none'
        for name in ('step', 'zero_grad', 'state_dict', 'load_state_dict'):
            if not callable(getattr(torch.optim.Adam, name, None)):
                ok, why = False, 'torch.optim.Adam.%s missing' % name
    except Exception as e:                                  # (no source available: a frozen build — trust the attribute checks alone)
        if not all(callable(getattr(torch.optim.Adam, n, None)) for n in ('step', 'zero_grad', 'state_dict', 'load_state_dict')):
            ok, why = False, repr(e)
    if not ok and not _WARNED[0]:
        _WARNED[0] = True
        warnings.warn('invr.optim: torch %s changed the optimizer internals the zero-edit adoption relies on (%s): the host\'s '
                      'torch.optim.Adam keeps its own step (dense table gradients, ~3x slower per iteration); use invr.driver.make_optimizer '
                      'or invr.optim.fuse() for the fused step' % (torch.__version__, why), RuntimeWarning)
    return ok


_WARNED = [False]
_SUPPORTED = [None]


def adopt(opt, net, attach=None):
    """Bind the host's torch.optim.Adam `opt` (over exactly `net`'s parameters) to the fused step; returns the inner FusedAdam."""
    d = opt.defaults
    inner = FusedAdam([{'params': [torch.zeros(1)]}], d['lr'], betas=d['betas'], eps=d['eps'], weight_decay=d['weight_decay'])
    inner.param_groups = opt.param_groups                   # the SAME list of the SAME dicts
    inner.state = opt.state                                 # the SAME state mapping (step / exp_avg / exp_avg_sq per parameter)
    if attach is None:
        import torch.distributed as dist
        attach = not (dist.is_available() and dist.is_initialized())
    if attach and hasattr(net, 'tpose_human') and getattr(net, 'cfg', {}).get('train_fused', True):
        inner.attach(net)
    orig_zero, orig_sd, orig_load = opt.zero_grad, opt.state_dict, opt.load_state_dict

    def step(self, closure=None):
        self._opt_called = True                             # (what torch's lr_scheduler wrapper of step() records)
        return inner.step(closure)

    def zero_grad(self, set_to_none=True):
        orig_zero(set_to_none)
        if inner.arena is not None:
            inner.arena.zero()

    def state_dict(self):
        inner._flush_steps()
        return orig_sd()

    def load_state_dict(self, sd):
        inner._flush_steps()
        orig_load(sd)
        inner.state, inner.param_groups = self.state, self.param_groups          # (load_state_dict rebinds both)
        inner._plan_key = None
    step._wrapped_by_lr_sched = True                        # (torch's lr_scheduler looks for its own step wrapper's mark, else it warns)
    opt.step = _types.MethodType(step, opt)
    opt.zero_grad = _types.MethodType(zero_grad, opt)
    opt.state_dict = _types.MethodType(state_dict, opt)
    opt.load_state_dict = _types.MethodType(load_state_dict, opt)
    opt._invr_inner = inner
    return inner


def _step_pre_hook(opt, args, kwargs):
    if not _ADOPT_NETS or type(opt) is not torch.optim.Adam or getattr(opt, '_invr_inner', None) is not None:
        return None
    # (torch hands the hook the wrapper's own positional arguments: args[0] is the optimizer, a positional closure is args[1])
    if (len(args) > 1 and args[1] is not None) or kwargs.get('closure') is not None:
        # step(closure): torch's own step runs the closure AFTER this hook — adopting now would attach the arena in front of a backward
        # whose table gradients this step then never applies.  Leave this step to torch; a later closure-free step adopts.
        return None
    if _SUPPORTED[0] is None:
        _SUPPORTED[0] = _adoption_supported()
    if not _SUPPORTED[0]:
        return None
    for net in list(_ADOPT_NETS):
        if _adam_matches(opt, net):
            inner = adopt(opt, net)
            inner.step()                                    # this step, from the dense gradients the first backward produced
            for g in opt.param_groups:                      # ... and torch's own step, which follows this hook, finds nothing to do
                for p in g['params']:
                    p.grad = None
            break
    return None


def adopt_on_first_step(net):
    """Register `net` for the adoption above (idempotent; a weak reference)."""
    if _os.environ.get('INVR_NO_OPTIM_HOOK', '0') == '1' or not getattr(net, 'cfg', {}).get('fused_optimizer_hook', True):
        return False
    _ADOPT_NETS.add(net)
    if _HOOK[0] is None:
        from torch.optim.optimizer import register_optimizer_step_pre_hook
        _HOOK[0] = register_optimizer_step_pre_hook(_step_pre_hook)
    return True
