"""Drop-in for lib/train/trainers/inb_trainer.py::NetworkWrapper (reference :19-248).

Same constructor / ``forward(batch, epoch=-1, split='train')`` signature and return tuple
``(ret, loss, scalar_stats, image_stats)``; the loss terms the INB configs produce are assembled as
the reference does (pair regulariser :45-48 with crit.reg_raw_crit, distortion :84-87, offset
:89-92, image loss :176-214 incl. the patch branch :188-214).

cfg.use_lpips (True in configs/inb/inb_377.yaml): the patch is re-assembled from `mask_at_box` exactly as the
reference does and handed to a perceptual-loss module — `NetworkWrapper(net, perceptual_loss=...)`, else the host
application's own `lib.train.trainers.loss.perceptual_loss.PerceptualLoss` when this wrapper runs inside the reference
(torchvision present), else `invr.losses.PerceptualLoss(weights=cfg.vgg19_weights)`.  With none of the three the
constructor RAISES: the wrapper never trains a different objective silently.  use_ssim / use_fourier / use_tv_image are
resolved from the host application the same way (they are off in every INB config) or rejected.
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


def _host_loss(module, cls):
    """A loss class of the hosting reference application (lib.train.trainers.loss.*), or None when not hosted."""
    import importlib
    import sys
    if 'lib.config' not in sys.modules:
        return None
    try:
        return getattr(importlib.import_module(module), cls)
    except Exception:
        return None


def assemble_patch(values, mask_at_box, H, W):
    """inb_trainer.py:196-203: `img = zeros(H,W,C); img[mask_at_box] = values` for values (1,Nr,C) / (Nr,C) of the rays
    inside the body AABB, in mask order.  Written as a gather by the running count of the mask, so that no boolean
    indexing (a host-syncing nonzero) is involved and the gradient flows to `values`."""
    v = values.reshape(-1, values.shape[-1])
    m = mask_at_box.reshape(-1).to(v.device).bool()
    idx = (torch.cumsum(m.to(torch.int64), 0) - 1).clamp_(min=0)
    if v.shape[0] == 0:
        return torch.zeros(H, W, v.shape[-1], device=v.device, dtype=v.dtype)
    img = v[idx.clamp_(max=v.shape[0] - 1)] * m[:, None].to(v.dtype)
    return img.reshape(H, W, v.shape[-1])


class NetworkWrapper(nn.Module):
    def __init__(self, net, perceptual_loss=None):
        super().__init__()
        self.net = net
        self.renderer = Renderer(self.net)
        self.cfg = cfg = getattr(net, 'cfg', global_cfg)
        # a torch.optim.Adam the host builds over this network's parameters (lib/train/optimizer.py:15-31) is bound to the fused
        # step at its first step() — invr.optim.adopt_on_first_step: train_net.py unchanged, FusedAdam's speed
        from .optim import adopt_on_first_step
        adopt_on_first_step(net)
        self.img2mse = lambda x, y: torch.mean((x - y) ** 2)
        if cfg.get('use_lpips', False):                                                  # inb_trainer.py:28-29
            if perceptual_loss is None:
                host = _host_loss('lib.train.trainers.loss.perceptual_loss', 'PerceptualLoss')
                if host is not None:
                    perceptual_loss = host()
                elif cfg.get('vgg19_weights', None):
                    from .losses import PerceptualLoss
                    perceptual_loss = PerceptualLoss(weights=cfg.vgg19_weights)
                else:
                    raise RuntimeError('cfg.use_lpips is set (configs/inb/inb_377.yaml:196) but no perceptual loss is available: pass '
                                       'NetworkWrapper(net, perceptual_loss=module), set cfg.vgg19_weights to a torchvision VGG19 '
                                       'state_dict, run inside the reference (torchvision), or set use_lpips False for the plain MSE')
            # the reference's PerceptualLoss moves its VGG to the GPU in its constructor (perceptual_loss.py:50); here the module
            # follows the network's device — the wrapper is normally built after net.to(device) and not moved again
            dev = next(net.parameters()).device
            if isinstance(perceptual_loss, torch.nn.Module):
                perceptual_loss = perceptual_loss.to(dev)
            self.perceptual_loss = perceptual_loss
        for flag, mod, cls, attr in (('use_ssim', 'lib.utils.loss_utils', 'SSIM', 'ssim_loss'),
                                     ('use_fourier', 'lib.train.trainers.loss.fourier_loss', 'FourierLoss', 'fourier_loss'),
                                     ('use_tv_image', 'lib.train.trainers.loss.tv_image_loss', 'TVImageLoss', 'tv_image_loss')):
            if cfg.get(flag, False):                                                     # :31-38, off in every INB config
                host = _host_loss(mod, cls)
                if host is None:
                    raise RuntimeError('cfg.%s is set but %s.%s is only available inside the reference application' % (flag, mod, cls))
                setattr(self, attr, host(window_size=11) if flag == 'use_ssim' else host())

    def _fused_objective(self, ret, batch):
        """split 'train' on the fused node with the plain MSE image term: the whole objective as ONE autograd node (autograd.TrainLossFn;
        same terms, same order of additions as the op-by-op form below) -> (loss, scalar_stats) or None when it does not apply."""
        cfg = self.cfg
        last = getattr(self.renderer, 'last_train', None)
        if (last is None or last.get('terms') is None or not cfg.get('train_fused_loss', True)
                or cfg.get('use_lpips', False) or cfg.get('use_ssim', False) or cfg.get('use_fourier', False) or cfg.get('use_tv_image', False)):
            return None
        from .autograd import TrainLossFn
        dist = ret['reg_distortion_loss'][0] if dict.__contains__(ret, 'reg_distortion_loss') else None
        out, err = TrainLossFn.apply(ret['rgb_map'][0], batch['rgb'][0], dist, last['terms'], float(cfg.pair_loss_weight), float(cfg.reg_dist_weight),
                                     float(cfg.resd_loss_weight), bool(cfg.use_pair_reg))
        stats = {'loss': out[0], 'img_loss': out[1].detach(), 'psnr': out[2:3].detach(), 'offset_loss': out[4].detach()}
        if dist is not None:
            stats['reg_dist'] = out[3].detach()
        if cfg.use_pair_reg:
            stats['pair_loss'] = out[5].detach()
        ret['error'] = err[None]
        return out[0], stats

    def forward(self, batch, epoch=-1, split='train'):
        cfg = self.cfg
        ret = self.renderer.render(batch, test=False, epoch=epoch)
        if split == 'train' and self.net.training:
            fused = self._fused_objective(ret, batch)
            if fused is not None:
                return ret, fused[0], fused[1], {}
        scalar_stats = {}
        dev = batch['latent_index'].device
        loss = torch.tensor(0.0, device=dev)
        if 'pair_loss' in ret:                      # fused path: crit.reg_raw_crit reduced on the device (0 when no row qualifies)
            scalar_stats['pair_loss'] = ret['pair_loss']
            loss = loss + cfg.pair_loss_weight * ret['pair_loss']
        elif 'oresd' in ret and ret['oresd'].numel():
            oresd = reg_raw_crit(ret['oresd'].to(dev))
            scalar_stats['pair_loss'] = oresd
            loss = loss + cfg.pair_loss_weight * oresd
        if 'reg_distortion_loss' in ret:
            rd = ret['reg_distortion_loss'].to(dev).mean()
            scalar_stats['reg_dist'] = rd
            loss = loss + cfg.reg_dist_weight * rd
        if 'offset_loss' in ret:                    # fused path: mean over the dense (Na*P) rows, reduced on the device
            scalar_stats['offset_loss'] = ret['offset_loss']
            loss = loss + cfg.resd_loss_weight * ret['offset_loss']
        elif 'resd' in ret:
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
            err = torch.abs(rgb_map - batch['rgb']).sum(dim=-1).detach()
            # the reference reads img_loss back per iteration for its psnr stat (.item()); here it stays a device scalar
            psnr = -10.0 * torch.log(img_loss.detach()) / np.log(10)
            scalar_stats.update({'img_loss': img_loss, 'psnr': psnr.reshape(1)})
            if cfg.get('use_lpips', False) or cfg.get('use_ssim', False) or cfg.get('use_fourier', False) or cfg.get('use_tv_image', False):
                H, W = int(batch['H'].item()), int(batch['W'].item())                   # :188-203: the rays of a patch back on its pixels
                img_pred = assemble_patch(rgb_map, batch['mask_at_box'][0], H, W)
                img_gt = assemble_patch(batch['rgb'], batch['mask_at_box'][0], H, W)
                if cfg.get('use_lpips', False):                                          # :206-209: NO separate MSE term
                    lp = self.perceptual_loss(img_pred.permute(2, 0, 1)[None], img_gt.permute(2, 0, 1)[None])
                    scalar_stats['lpips_loss'] = lp
                    loss = loss + lp
                elif cfg.get('use_ssim', False):
                    ss = 1 - self.ssim_loss(img_pred.permute(2, 0, 1)[None], img_gt.permute(2, 0, 1)[None])
                    scalar_stats['ssim_loss'] = ss
                    loss = loss + 0.1 * ss + img_loss
                elif cfg.get('use_fourier', False):
                    fl = self.fourier_loss(img_pred, img_gt)
                    scalar_stats['fourier_loss'] = fl
                    loss = loss + 0.1 * fl + img_loss
                else:
                    mask_gt = assemble_patch(batch['occupancy'].to(rgb_map.dtype)[..., None], batch['mask_at_box'][0], H, W)[..., 0] > 0
                    tv = self.tv_image_loss(img_pred, img_gt, mask_gt)
                    scalar_stats['tv_loss'] = tv
                    loss = loss + 0.01 * tv + img_loss
            else:
                loss = loss + img_loss
            scalar_stats['loss'] = loss
            ret['error'] = err
        else:
            raise NotImplementedError
        return ret, loss, scalar_stats, image_stats
