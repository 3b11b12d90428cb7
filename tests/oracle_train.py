"""Test infrastructure: the training objective of the reference (NetworkWrapper.forward with the INB configs,
lib/train/trainers/inb_trainer.py:40-98,176-214 with use_lpips False; Renderer train branches inb_renderer.py:78-103;
crit.reg_raw_crit crit.py:8-18) evaluated with the ORACLE (oracle/nvr_oracle.py, CPU torch ops) under torch autograd.
Checker only — never imported by the product."""
import torch

from oracle import nvr_oracle as O


def reg_raw_crit(x):
    n = x.shape[1] // 2
    v = x / (x.norm(dim=-1, keepdim=True) + 1e-8)
    return (v[:, n:] - v[:, :n]).norm(dim=-1).mean()


def train_loss(sd, cfg, batch, jitter, noise_dense, chunk=64):
    """sd: reference-keyed dict of CPU tensors (leaves that require grad get gradients); batch: collated CPU batch;
    jitter (n_rays,S) uniform; noise_dense (N*5,3) uniform per dense (survivor slot, part) row -> (loss, stats dict)."""
    model = O.Model(sd, cfg)
    S = cfg.N_samples
    ret = O.render(model, batch, n_samples=S, jitter=None if jitter is None else jitter[None], want_train=True, chunk=chunk)
    img = ((ret['rgb_map'] - batch['rgb']) ** 2).mean()
    loss = img
    stats = {'img_loss': img}
    resd, tocc, tpts = ret['resd'], ret['tocc'], ret['tpts']                # (Na,P,3), (Na,P), (Na,P,3)
    if cfg.use_reg_distortion:
        d = O.distortion_loss(ret['weights'], ret['z']).mean()
        stats['reg_dist'] = d
        loss = loss + cfg.reg_dist_weight * d
    off = torch.norm(resd.reshape(1, -1, 3), dim=2).mean()
    stats['offset_loss'] = off
    loss = loss + cfg.resd_loss_weight * off
    if cfg.use_pair_reg:
        reg = ((tocc.detach().reshape(-1) - 0.5).abs() < 0.02).nonzero(as_tuple=True)[0]
        if reg.numel():
            b = {k: (v[0] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in batch.items()}
            nb = tpts.detach().reshape(-1, 3)[reg] + (noise_dense[reg] - 0.5) * 0.01
            nei = O.deformer(nb, sd, model.dspec, b['tuv'], b['tbounds'], b['frame_dim'])
            pair = reg_raw_crit(torch.cat([resd.reshape(-1, 3)[reg][None], nei[None]], 1))
            stats['pair_loss'] = pair
            loss = loss + cfg.pair_loss_weight * pair
    stats['loss'] = loss
    return loss, stats


def adopt_batch_bounds(sd, cfg, batch):
    """part_base_embedder.py:107-109 at iter_step == 1: the part grids' bounds parameter is re-created from the batch."""
    from invr.config import PART_NAMES
    for i, name in enumerate(PART_NAMES):
        if cfg.partnet[name].embedder.kwargs.get('use_batch_bounds', False):
            sd['tpose_human.part_networks.%d.embedder.bounds' % i] = batch['bounds'][0][i].detach().clone()
