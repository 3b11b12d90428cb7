"""Row f1: the optimiser of the reference's training loop as one HIP launch.

`FusedAdam` is torch.optim.Adam (lib/train/optimizer.py:13-31 builds it with one parameter group per tensor,
`eps=cfg.train.eps`) with the update of every tensor done by `invr_adam_step`; hyper-parameters, per-group `lr`
(the reference's schedulers write `group['lr']`), `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq`) and the
skip-tensors-without-gradient rule are torch's, so optimiser checkpoints interchange with the reference's."""
import ctypes as C
import math

import numpy as np
import torch

from . import _abi


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plan_key = None

    def _plan(self, entries):
        """Device chunk tables for the tensors that have a gradient this step (cached while the set is unchanged)."""
        key = tuple((p.data_ptr(), p.numel()) for p, _ in entries)
        if key != self._plan_key:
            E = _abi.lib().invr_adam_chunk_elems()
            ct, ci = [], []
            for t, (p, _) in enumerate(entries):
                n = (p.numel() + E - 1) // E
                ct.append(np.full(n, t, np.int32))
                ci.append(np.arange(n, dtype=np.int32))
            dev = entries[0][0].device
            self._chunk_tensor = torch.from_numpy(np.concatenate(ct)).to(dev)
            self._chunk_index = torch.from_numpy(np.concatenate(ci)).to(dev)
            self._plan_key = key
        return self._chunk_tensor, self._chunk_index

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries = []
        betas = eps = None
        for group in self.param_groups:
            if betas is None:
                betas, eps = group['betas'], group['eps']
            assert (betas, eps) == (group['betas'], group['eps']), 'FusedAdam: betas / eps must be the same in all groups'
            for p in group['params']:
                if p.grad is None:
                    continue
                assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse
                st = self.state[p]
                if not st:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                entries.append((p, group))
        if not entries:
            return loss
        tab = (_abi.InvrAdamTensor * len(entries))()
        keep = []
        for e, (p, group) in zip(tab, entries):
            st = self.state[p]
            g = p.grad.contiguous()
            keep.append(g)
            k = float(st['step'])
            e.param, e.grad, e.exp_avg, e.exp_avg_sq = p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
            e.numel, e.lr, e.weight_decay = p.numel(), group['lr'], group['weight_decay']
            e.bc1, e.bc2_sqrt = 1.0 - betas[0] ** k, math.sqrt(1.0 - betas[1] ** k)
        dev = entries[0][0].device
        host = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8)
        table = host.to(dev, non_blocking=False)
        ct, ci = self._plan(entries)
        _abi.check(_abi.lib().invr_adam_step(C.c_void_p(table.data_ptr()), _abi.ptr(ct, torch.int32), _abi.ptr(ci, torch.int32),
                                             ct.numel(), betas[0], betas[1], eps, _abi.stream_ptr()))
        for p, _ in entries:          # the kernel wrote through raw pointers: tell autograd / version-keyed caches (Embedder.row_sums)
            torch.autograd.graph.increment_version(p)
            torch.autograd.graph.increment_version(self.state[p]['exp_avg'])
            torch.autograd.graph.increment_version(self.state[p]['exp_avg_sq'])
        return loss
