"""Drop-in for lib/train/trainers/inb_trainer.py::NetworkWrapper (reference :19-248).

Same constructor / ``forward(batch, epoch=-1, split='train')`` signature and return tuple
``(ret, loss, scalar_stats, image_stats)``; the loss terms the INB configs produce are assembled as
the reference does (pair regulariser :45-48 with crit.reg_raw_crit, distortion :84-87, offset
:89-92, image loss :176-214).  The LPIPS branch needs torchvision's VGG19 (absent on this image):
with cfg.use_lpips the plain MSE is used and ``scalar_stats['lpips_loss']`` is not produced.
"""
import numpy as np
import torch
import torch.nn as nn

from .config import cfg as global_cfg
from .renderer import Renderer


def reg_raw_crit(x):
    """lib/train/trainers/crit.py:8-18."""
    n_pts = x.shape[1] // 2
    length = x.norm(dim=-1, keepdim=True)
    vector = x / (length + 1e-8)
    return (vector[:, n_pts:, :] - vector[:, :n_pts, :]).norm(dim=-1).mean()


class NetworkWrapper(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.renderer = Renderer(self.net)
        self.cfg = getattr(net, 'cfg', global_cfg)
        self.img2mse = lambda x, y: torch.mean((x - y) ** 2)

    def forward(self, batch, epoch=-1, split='train'):
        cfg = self.cfg
        ret = self.renderer.render(batch, test=False, epoch=epoch)
        scalar_stats = {}
        dev = batch['latent_index'].device
        loss = torch.tensor(0.0, device=dev)
        if 'oresd' in ret and ret['oresd'].numel():
            oresd = reg_raw_crit(ret['oresd'].to(dev))
            scalar_stats['pair_loss'] = oresd
            loss = loss + cfg.pair_loss_weight * oresd
        if 'reg_distortion_loss' in ret:
            rd = ret['reg_distortion_loss'].to(dev).mean()
            scalar_stats['reg_dist'] = rd
            loss = loss + cfg.reg_dist_weight * rd
        if 'resd' in ret:
            off = torch.norm(ret['resd'].to(dev), dim=2).mean()
            scalar_stats['offset_loss'] = off
            loss = loss + cfg.resd_loss_weight * off
        image_stats = {}
        if split == 'val':
            rgb_pred = ret['rgb_map'][0].detach().cpu()
            rgb_gt = batch['rgb'][0].detach().cpu()
            mask = batch['mask_at_box'][0].detach().cpu()
            H, W = int(batch['H'].item()), int(batch['W'].item())
            mask = mask.reshape(H, W)
            img_pred = torch.zeros((H, W, 3)); img_pred[mask] = rgb_pred
            img_gt = torch.zeros((H, W, 3)); img_gt[mask] = rgb_gt
            scalar_stats['loss'] = loss
            image_stats = {'img_gt': img_gt, 'img_pred': img_pred, 'error_map': torch.abs(img_pred - img_gt).sum(-1)}
        elif split == 'train':
            rgb_map = ret['rgb_map'].to(dev)
            img_loss = self.img2mse(rgb_map, batch['rgb'])
            err = torch.abs(rgb_map - batch['rgb']).sum(dim=-1).detach().cpu()
            psnr = -10 * np.log(img_loss.item()) / np.log(10)
            scalar_stats.update({'img_loss': img_loss, 'psnr': torch.Tensor([psnr])})
            loss = loss + img_loss
            scalar_stats['loss'] = loss
            ret['error'] = err
        else:
            raise NotImplementedError
        return ret, loss, scalar_stats, image_stats
