"""Tensor-level wrappers of libinvr's stage entry points (include/invr.h "stage-level entry points").

Each function is one C-ABI call on the current torch stream; inputs are device tensors, `scene` / `model` are the
ctypes structs of a RenderContext (Network.prepare).  Used by the training path and by the parity tests, which pin
the render pipeline's production kernels against these brute-force / dense formulations of the same reference
functions.
"""
import ctypes as C

import torch

from . import _abi
from .config import NUM_PARTS


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def pose_points(scene, ray_o, ray_d, near, far, n_samples, sample_idx=None, jitter=None, want_dirs=True):
    """get_wsampling_points + world->pose (inb_renderer.py:15-31, blend_utils.py:366-382) of selected ray-samples
    (sample_idx int32 = ray*S + s; None = all, in order) -> pose_pts (n,3), pose_dirs (n,3) — bit-identical to
    what the render kernels compute (same device function)."""
    ray_o, ray_d, near, far = _f32(ray_o), _f32(ray_d), _f32(near), _f32(far)
    R, S = ray_o.shape[0], int(n_samples)
    if sample_idx is not None:
        sample_idx = sample_idx.to(torch.int32).contiguous()
        n = sample_idx.numel()
    else:
        n = R * S
    dev = ray_o.device
    pts = torch.empty(n, 3, device=dev)
    dirs = torch.empty(n, 3, device=dev) if want_dirs else None
    if jitter is not None:
        jitter = _f32(jitter)
    _abi.check(_abi.lib().invr_pose_points(C.byref(scene), _abi.ptr(ray_o), _abi.ptr(ray_d), _abi.ptr(near), _abi.ptr(far),
                                           _abi.ptr(jitter), R, S, _abi.ptr(sample_idx, torch.int32), n, _abi.ptr(pts),
                                           _abi.ptr(dirs), _abi.stream_ptr()))
    return pts, dirs


def knn_neighbors(scene, pose_pts):
    """Brute-force per-part 4-NN (blend_utils.py:732-763): -> nn (n,P,4) int32, d2 (n,P,4), w (n,P,4), dist (n,P)."""
    x = _f32(pose_pts)
    n, dev = x.shape[0], x.device
    nn = torch.empty(n, NUM_PARTS, 4, dtype=torch.int32, device=dev)
    d2 = torch.empty(n, NUM_PARTS, 4, device=dev)
    w = torch.empty(n, NUM_PARTS, 4, device=dev)
    dist = torch.empty(n, NUM_PARTS, device=dev)
    _abi.check(_abi.lib().invr_knn_neighbors(C.byref(scene), _abi.ptr(x), n, _abi.ptr(nn, torch.int32), _abi.ptr(d2), _abi.ptr(w),
                                             _abi.ptr(dist), _abi.stream_ptr()))
    return nn, d2, w, dist


def knn_blend(scene, pose_pts):
    """pts_knn_blend_weights_multiassign_batch (blend_utils.py:817-825): -> bw (n,P,24), dist (n,P)."""
    x = _f32(pose_pts)
    n, dev = x.shape[0], x.device
    bw = torch.empty(n, NUM_PARTS, 24, device=dev)
    dist = torch.empty(n, NUM_PARTS, device=dev)
    _abi.check(_abi.lib().invr_knn_blend(C.byref(scene), _abi.ptr(x), n, _abi.ptr(bw), _abi.ptr(dist), _abi.stream_ptr()))
    return bw, dist


def warp_deform(scene, model, pose_pts, pose_dirs, bw, flag):
    """Network.pose_points_to_tpose_points (inb_part_network_multiassign.py:77-120), dense over all (point, part) pairs:
    -> tpose (n,P,3), tdirs (n,P,3), resd (n,P,3) (zeros where !flag)."""
    x, d, bw = _f32(pose_pts), _f32(pose_dirs), _f32(bw)
    flag = flag.to(torch.uint8).contiguous()
    n, dev = x.shape[0], x.device
    tp = torch.empty(n, NUM_PARTS, 3, device=dev)
    td = torch.empty(n, NUM_PARTS, 3, device=dev)
    rs = torch.empty(n, NUM_PARTS, 3, device=dev)
    _abi.check(_abi.lib().invr_warp_deform(C.byref(scene), C.byref(model), _abi.ptr(x), _abi.ptr(d), _abi.ptr(bw),
                                           _abi.ptr(flag, torch.uint8), n, _abi.ptr(tp), _abi.ptr(td), _abi.ptr(rs),
                                           _abi.stream_ptr()))
    return tp, td, rs
