"""Data-parallel training of the path (BASELINE configs[4]; reference: DistributedDataParallel around NetworkWrapper,
lib/train/trainers/trainer.py:21-26, one image / patch per rank through DistributedSampler, lib/datasets/samplers.py:75-132).

One process per GPU (torch.distributed; "nccl" = RCCL over xGMI, "gloo" in tests), full model replica per rank, each rank
renders and differentiates its own patch; gradients are AVERAGED over the ranks (DDP's semantics) before the optimiser step.
What is exchanged is the fused path's gradient arena (autograd.GradArena):

  * per part grid ONE compact row-scalar gradient array (invr_train_bwd) — 68 MB for the five inb_377 grids instead of the
    1.09 GB of dense table gradients DDP would all-reduce (the gradient of a sum-over-features table is one scalar per row, so
    nothing is lost): ring all-reduce time per iteration ~ 2 * 7/8 * 68 MB / 153 GB/s = 0.8 ms instead of 13 ms;
  * one flat buffer with every small tensor (MLPs, latent codes, deformer tables; 0.4 MB).

Overlap: TrainRenderFn.backward runs invr_train_bwd stage by stage and calls `reduce_part(p)` right after part p's kernels are
enqueued; the collective is asynchronous (its own stream, ordered after the producing kernels by the process group) and runs
beside the remaining backward stages; `wait()` (FusedAdam.step via optimizer hook, or explicitly) joins before the update.
"""
import torch
import torch.distributed as dist

from .dist import FORCE_COLLECTIVES


class GradReducer:
    def __init__(self, arena, group=None):
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        # largest gradient block first: its all-reduce overlaps the most remaining work
        sizes = [e.row_grad().numel() for e in arena.embedders]
        self.part_order = sorted(range(len(sizes)), key=lambda p: -sizes[p])
        self.pending = []
        arena.reducer = self

    def _all_reduce_mean(self, t):
        if self.world == 1 and not (dist.is_initialized() and FORCE_COLLECTIVES()):
            return
        if self.backend == 'nccl':
            self.pending.append(dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:                                   # gloo (tests): no AVG, device tensors go through the host
            if t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t.copy_(h.div_(self.world))
            else:
                dist.all_reduce(t, group=self.group)
                t.div_(self.world)

    def reduce_part(self, p):
        self._all_reduce_mean(self.arena.embedders[p].row_grad())

    def reduce_small(self):
        self._all_reduce_mean(self.arena.flat)

    def wait(self):
        """Order the current stream behind every outstanding all-reduce (no host synchronisation with nccl)."""
        for w in self.pending:
            w.wait()
        self.pending = []


def attach(optimizer, group=None):
    """Make a FusedAdam with a gradient arena data-parallel: gradients are averaged over the ranks of `group` during backward,
    `optimizer.step()` first joins the collectives."""
    assert optimizer.arena is not None, 'FusedAdam.attach(net) first'
    red = GradReducer(optimizer.arena, group)
    optimizer.register_step_pre_hook(lambda opt, args, kwargs: red.wait())
    return red


def broadcast_parameters(net, src=0, group=None):
    """Same initial replica on every rank (DDP does this at construction)."""
    for p in net.parameters():
        if dist.get_backend(group) == 'gloo' and p.is_cuda:
            h = p.detach().cpu()
            dist.broadcast(h, src, group=group)
            p.data.copy_(h)
        else:
            dist.broadcast(p.data, src, group=group)
