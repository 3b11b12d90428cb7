"""Minimal drivers around the drop-in classes — the two production callers of the path, restated so
they run where the reference is absent (the GPU box):

  run_evaluate  : the eval loop of run.py:61-90 (batch -> device, no_grad, renderer.render, metrics)
  train_step    : one optimisation step of Trainer.train (lib/train/trainers/trainer.py:108-149)
  psnr_metric   : Evaluator.psnr_metric / evaluate (lib/evaluators/if_nerf.py:28-31, 80-115): PSNR over
                  the whole HxW image with zeros outside mask_at_box
  save_model / load_network : the reference's .pth layout {net, optim, scheduler, recorder, epoch}
                  (lib/utils/net_utils.py:461-528) so checkpoints interchange in both directions
"""
import os

import numpy as np
import torch

from .renderer import Renderer


def psnr_metric(img_pred, img_gt):
    mse = np.mean((img_pred - img_gt) ** 2)
    return -10 * np.log(mse) / np.log(10)


def assemble_image(values, batch):
    """(n_rays,3) values of the rays inside the body AABB -> (H,W,3) image, zeros elsewhere (:84-91)."""
    mask = batch['mask_at_box'][0].detach().cpu().numpy().reshape(int(batch['H'].item()), int(batch['W'].item()))
    img = np.zeros(mask.shape + (3,))
    img[mask] = values
    return img


def _lanes_that_fit(want, device):
    """Frames in flight the free device memory carries: a lane holds a workspace (~5 GB for a 512x512x128 frame at the survivor
    bound Renderer keeps) and a raw buffer (0.5 GB); leave half of what is free to the caller."""
    try:
        free, _ = torch.cuda.mem_get_info(torch.device(device))
    except Exception:
        return want
    return max(1, min(want, int(free * 0.5 // (6 << 30))))


def run_evaluate(net, batches, device='cuda', in_flight=None, renderer=None, keep_maps=False):
    """-> dict(psnr=[...], mse=[...]) over an iterable of collated batches (CPU or device tensors): the loop of run.py:61-90 with
    `in_flight` frames kept on the GPU — frame f is submitted (Renderer.render returns at once, Renderer.in_flight lanes) and the
    metrics of frame f - in_flight + 1 are computed while the younger frames render; the values are those of one frame at a time
    (same kernels, a workspace per lane).  in_flight = 1: strictly render -> metrics -> next batch, as the reference.
    in_flight = None: cfg.render_in_flight when the config sets it, else 8 — capped by the free device memory.  A renderer the caller
    passed in gets its own in_flight back on return and its lanes' workspaces / raw buffers are released.
    keep_maps: also return the host rgb_map of every frame (tests)."""
    from collections import deque
    net.eval()
    renderer = renderer or Renderer(net)
    if in_flight is None:
        cfg_v = getattr(net, 'cfg', {}).get('render_in_flight', None) if hasattr(getattr(net, 'cfg', None), 'get') else None
        in_flight = int(cfg_v) if cfg_v else 8
    prev_in_flight = renderer.in_flight
    if torch.device(device).type == 'cuda' and torch.cuda.is_available():
        in_flight = _lanes_that_fit(max(1, int(in_flight)), device)
    renderer.in_flight = max(1, int(in_flight))
    out = {'psnr': [], 'mse': []}
    if keep_maps:
        out['rgb_map'] = []
    queue = deque()

    def finish(ret, batch):
        rgb = ret['rgb_map'][0].detach().cpu()
        pred = assemble_image(rgb.numpy(), batch)
        gt = assemble_image(batch['rgb'][0].detach().cpu().numpy(), batch)
        out['mse'].append(float(np.mean((pred - gt) ** 2)))
        out['psnr'].append(float(psnr_metric(pred.reshape(-1, 3), gt.reshape(-1, 3))))
        if keep_maps:
            out['rgb_map'].append(rgb)

    try:
        for batch in batches:
            batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
            with torch.no_grad():
                ret = renderer.render(batch)
            queue.append((ret, batch))
            while len(queue) >= renderer.in_flight:
                finish(*queue.popleft())
        while queue:
            finish(*queue.popleft())
    finally:
        renderer.in_flight = prev_in_flight
        if hasattr(renderer, 'flush'):
            renderer.flush(release=True)
    return out


_SCALER = None


def train_step(wrapper, optimizer, batch, iter_step, epoch=0, scaler=None):
    """One optimisation step in the reference's own form (lib/train/trainers/trainer.py:108-149): add_iter_step, the wrapper
    forward under autocast(enabled=cfg.use_amp), loss.mean(), zero_grad(set_to_none), scaler.scale(loss).backward(),
    scaler.step(optimizer), scaler.update() with GradScaler(enabled=cfg.use_amp) — use_amp is False in every INB config and
    rejected by invr.config.validate, so autocast and the scaler are the disabled pass-throughs the reference runs with.
    Returns (loss value, scalar_stats)."""
    global _SCALER
    if scaler is None:
        if _SCALER is None:
            _SCALER = torch.amp.GradScaler('cuda', enabled=False)
        scaler = _SCALER
    batch['iter_step'] = iter_step
    with torch.amp.autocast('cuda', enabled=False):
        ret, loss, stats, _ = wrapper(batch, epoch, split='train')
    loss = loss.mean()
    optimizer.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(optimizer)
    scaler.update()
    return loss.detach(), stats          # a device scalar: reading it (float()) is the caller's synchronisation point


def make_optimizer(net, lr=5e-4, eps=1e-15, weight_decay=0.0, fused=None):
    """lib/train/optimizer.py:15-31: Adam, one parameter group per tensor; keys without 'data' in their name (every key of this
    network) take lr * cfg.mlp_weight_decay (1.0 in lib/config/config.py:245 and in every INB yaml).  fused=False builds exactly the
    reference's optimizer (torch.optim.Adam over the same groups): the fused training node then hands it dense table gradients."""
    mwd = float(getattr(net, 'cfg', {}).get('mlp_weight_decay', 1.0))
    groups = [{'params': [p], 'lr': lr if 'data' in k else lr * mwd, 'weight_decay': weight_decay}
              for k, p in net.named_parameters() if p.requires_grad]
    if fused is None:
        fused = all(p.is_cuda for g in groups for p in g['params'])
    if fused:                      # one HIP launch for all tensors (invr_adam_step) instead of ~8 launches per tensor
        from .optim import FusedAdam
        opt = FusedAdam(groups, lr, eps=eps, weight_decay=weight_decay)
        if hasattr(net, 'tpose_human') and getattr(net, 'cfg', {}).get('train_fused', True):
            opt.attach(net)        # persistent gradient arena + row-scalar table gradients for the fused training path
        return opt
    return torch.optim.Adam(groups, lr, eps=eps, weight_decay=weight_decay)


def save_model(net, optim, scheduler, recorder, model_dir, epoch, last=False):
    os.makedirs(model_dir, exist_ok=True)
    sd = lambda x: x.state_dict() if x is not None else {}
    model = {'net': net.state_dict(), 'optim': sd(optim), 'scheduler': sd(scheduler), 'recorder': sd(recorder), 'epoch': epoch}
    torch.save(model, os.path.join(model_dir, 'latest.pth' if last else '{}.pth'.format(epoch)))


def load_network(net, model_dir, epoch=-1, strict=True):
    """net_utils.load_network: latest.pth, else the highest epoch; returns the next epoch (0 if nothing)."""
    if not os.path.exists(model_dir):
        return 0
    if os.path.isdir(model_dir):
        names = os.listdir(model_dir)
        pths = [int(p.split('.')[0]) for p in names if p != 'latest.pth' and p.endswith('.pth')]
        if not pths and 'latest.pth' not in names:
            return 0
        pth = ('latest' if 'latest.pth' in names else max(pths)) if epoch == -1 else epoch
        path = os.path.join(model_dir, '{}.pth'.format(pth))
    else:
        path = model_dir
    ck = torch.load(path, map_location='cpu')
    net.load_state_dict(ck['net'], strict=strict)
    return ck['epoch'] + 1


class ExponentialLR(torch.optim.lr_scheduler._LRScheduler):
    """lib/utils/optimizer/lr_scheduler.py:66-75 (cfg.train.scheduler type "exponential"): lr = base_lr * gamma ** (epoch /
    decay_epochs), stepped once per epoch (train_net.py:138)."""

    def __init__(self, optimizer, decay_epochs, gamma=0.1, last_epoch=-1):
        self.decay_epochs, self.gamma = decay_epochs, gamma
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [base_lr * self.gamma ** (self.last_epoch / self.decay_epochs) for base_lr in self.base_lrs]


def change_training_stages(cfg, epoch, stages):
    """train_net.py:64-75: the last stage whose `_start` <= epoch writes its keys into cfg."""
    for stage in (stages or [])[::-1]:
        if epoch >= stage['_start']:
            for k, v in stage.items():
                if k != '_start':
                    cfg[k] = v
            break


def train(wrapper, optimizer, batch_fn, epochs, ep_iter, scheduler=None, stages=None, budget_s=None, on_step=None):
    """The reference's training schedule (train_net.py:131-158 around Trainer.train, trainer.py:64-185) on the drop-in classes:
    per epoch change_training_stages, `ep_iter` iterations with iter_step = index + 1 (so the part grids adopt the batch's
    bounds at the first iteration of every epoch, part_base_embedder.py:107-109), scheduler.step() after the epoch.
    `batch_fn(epoch, index)` -> collated batch dict on the device.  Stops when `budget_s` seconds of wall clock are spent
    (BASELINE configs[3]: "5-min budget").  Nothing in the loop reads a device value back: the loss history is a list of
    device scalars, one synchronisation at the end.  Returns dict(iterations, seconds, ray_samples, losses)."""
    import time
    cfg = wrapper.cfg
    losses, n_samples, it = [], 0, 0
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    t0 = time.perf_counter()
    done = False
    for epoch in range(epochs):
        change_training_stages(cfg, epoch, stages)
        for index in range(ep_iter):
            batch = batch_fn(epoch, index)
            loss, stats = train_step(wrapper, optimizer, batch, index + 1, epoch)
            losses.append(loss)
            n_samples += int(batch['ray_o'].shape[1]) * int(cfg.N_samples)
            it += 1
            if on_step is not None:
                on_step(epoch, index, loss, stats)
            if budget_s is not None and (it & 15) == 0 and time.perf_counter() - t0 > budget_s:
                done = True
                break
        if scheduler is not None:
            scheduler.step()
        if done:
            break
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    dt = time.perf_counter() - t0
    return {'iterations': it, 'seconds': dt, 'ray_samples': n_samples, 'losses': [float(l) for l in losses]}
