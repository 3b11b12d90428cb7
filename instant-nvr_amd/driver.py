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


def run_evaluate(net, batches, device='cuda'):
    """-> dict(psnr=[...], mse=[...]) over an iterable of collated batches (CPU or device tensors)."""
    net.eval()
    renderer = Renderer(net)
    out = {'psnr': [], 'mse': []}
    for batch in batches:
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        with torch.no_grad():
            ret = renderer.render(batch)
        pred = assemble_image(ret['rgb_map'][0].detach().cpu().numpy(), batch)
        gt = assemble_image(batch['rgb'][0].detach().cpu().numpy(), batch)
        out['mse'].append(float(np.mean((pred - gt) ** 2)))
        out['psnr'].append(float(psnr_metric(pred.reshape(-1, 3), gt.reshape(-1, 3))))
    return out


def train_step(wrapper, optimizer, batch, iter_step, epoch=0):
    """trainer.py:108-149 without AMP: add_iter_step, wrapper forward, loss.mean(), zero_grad(set_to_none),
    backward, step.  Returns (loss value, scalar_stats)."""
    batch['iter_step'] = iter_step
    ret, loss, stats, _ = wrapper(batch, epoch, split='train')
    loss = loss.mean()
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    return loss.detach(), stats          # a device scalar: reading it (float()) is the caller's synchronisation point


def make_optimizer(net, lr=5e-4, eps=1e-15, weight_decay=0.0, fused=None):
    """lib/train/optimizer.py:15-31: Adam, one parameter group per tensor."""
    groups = [{'params': [p], 'lr': lr, 'weight_decay': weight_decay} for p in net.parameters() if p.requires_grad]
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
