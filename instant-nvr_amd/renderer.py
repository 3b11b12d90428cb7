"""Drop-in for lib/networks/renderer/inb_renderer.py::Renderer (reference :11-239).

Same constructor and ``render(batch, test=False, epoch=-1)`` signature and the same output
dict; the per-chunk Python loop, ``get_wsampling_points``, ``get_density_color``, the network
forward and ``volume_rendering`` are one stream-ordered libinvr call over the whole ray list
(chunk-free: HBM holds the full frame's intermediates).
"""
import torch

from .config import cfg as global_cfg

MAX_SAMPLES_PER_CALL = (1 << 31) - 1


class Renderer:
    def __init__(self, net):
        self.net = net
        self.cfg = getattr(net, 'cfg', global_cfg)
        # knobs that are not part of the reference signature
        self.eval_to_cpu = True        # reference moves every eval output to the CPU (:199-200)
        self.want_raw = True           # reference always returns raw/occ

    def render(self, batch, test=False, epoch=-1):
        cfg = self.cfg
        ray_o, ray_d = batch['ray_o'], batch['ray_d']
        near, far = batch['near'], batch['far']
        if epoch != -1:
            batch['epoch'] = epoch
        n_batch, n_pixel = ray_o.shape[:2]
        assert n_batch == 1, 'the path asserts a batch of one frame (inb_part_network_multiassign.py:84,155)'
        S = int(cfg.N_samples)
        training = self.net.training
        jitter = None
        if cfg.perturb > 0. and training:
            jitter = torch.rand((n_pixel, S), device=ray_o.device, dtype=torch.float32)    # :24
        if training:
            raise NotImplementedError('train-mode render (backward kernels) is the next row of SURVEY.md §8(f)')
        per_call = max(1, MAX_SAMPLES_PER_CALL // S)
        outs = []
        for i in range(0, n_pixel, per_call):
            sl = slice(i, i + per_call)
            outs.append(self.net.render_rays(batch, ray_o[0, sl], ray_d[0, sl], near[0, sl], far[0, sl], S,
                                             jitter=None if jitter is None else jitter[sl], want_raw=self.want_raw))
        cat = (lambda k: outs[0][k]) if len(outs) == 1 else (lambda k: torch.cat([o[k] for o in outs], 0))
        ret = {'rgb_map': cat('rgb_map')[None], 'acc_map': cat('acc_map')[None]}
        if self.want_raw:
            ret['raw'] = cat('raw')[None]
            ret['occ'] = cat('occ')[None, :, None]
        self.last_stats = outs[-1]['stats']
        if self.eval_to_cpu:
            ret = {k: v.detach().cpu() for k, v in ret.items()}
        return ret
