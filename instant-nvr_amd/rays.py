"""Row f2 of SURVEY.md §8: get_rays_within_bounds (lib/utils/if_nerf/if_nerf_data_utils.py:313-327) on the
device — removes the per-frame 262k-ray NumPy pass and H2D copy of a full-image render."""
import ctypes as C

import numpy as np
import torch

from . import _abi


def rays_within_bounds(H, W, K, R, T, bounds, device='cuda'):
    """K (3,3), R (3,3), T (3,1) float64 camera; bounds (2,3) float32 world AABB.
    -> ray_o, ray_d (n,3), near, far (n,), mask_at_box (H,W) bool — device tensors, rays inside the box
    only and in row-major pixel order, exactly like the reference."""
    K, R, T = np.asarray(K, np.float64), np.asarray(R, np.float64), np.asarray(T, np.float64)
    o = -np.dot(R.T, T).ravel()                               # get_rays: rays_o (:26)
    kinv = np.ascontiguousarray(np.linalg.inv(K))
    b = np.ascontiguousarray(np.asarray(bounds, np.float32).reshape(6))
    dp = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.POINTER(C.c_double))
    n = H * W
    ray_d = torch.empty(n, 3, device=device)
    near = torch.empty(n, device=device)
    far = torch.empty(n, device=device)
    mask = torch.empty(n, dtype=torch.uint8, device=device)
    Rc, Tc = np.ascontiguousarray(R), np.ascontiguousarray(T.ravel())
    _abi.check(_abi.lib().invr_generate_rays(dp(kinv), dp(Rc), dp(Tc), dp(o), b.ctypes.data_as(C.POINTER(C.c_float)), H, W,
                                             _abi.ptr(ray_d), _abi.ptr(near), _abi.ptr(far), _abi.ptr(mask, torch.uint8),
                                             _abi.stream_ptr()))
    m = mask.bool()
    ray_o = torch.from_numpy(o.astype(np.float32)).to(device)[None].expand(int(m.sum()), 3).contiguous()
    return ray_o, ray_d[m], near[m], far[m], m.view(H, W)
