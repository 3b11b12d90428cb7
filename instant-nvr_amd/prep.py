"""Row f4 of SURVEY.md §8: per-frame scene tensors on the device (no CPU dataset in the loop)."""
import torch

from . import _abi
from .config import NUM_PARTS


def rigid_transformation(poses, joints, parents):
    """get_rigid_transformation (if_nerf_data_utils.py:545-577): poses, joints (24,3), parents (24) -> A (24,4,4) f32."""
    dev = poses.device
    p = poses.to(torch.float64).contiguous()
    j = joints.to(torch.float64).contiguous()
    par = parents.to(torch.int32).contiguous()
    A = torch.empty(24, 4, 4, device=dev)
    _abi.check(_abi.lib().invr_rigid_transformation(_abi.ptr(p, torch.float64), _abi.ptr(j, torch.float64),
                                                    _abi.ptr(par, torch.int32), _abi.ptr(A), _abi.stream_ptr()))
    return A


def pack_parts(ppts, weights, parts, tpose, bbox_overlap=0.2):
    """tpose_dataset.py:570-600 -> part_pts (5,M,3), part_pbw (5,M,24), lengths2 (5) int64, bounds (5,2,3)."""
    dev = ppts.device
    V, W = weights.shape
    f = lambda t: t.to(torch.float32).contiguous()
    part_pts = torch.empty(NUM_PARTS, V, 3, device=dev)
    part_pbw = torch.empty(NUM_PARTS, V, W, device=dev)
    lengths2 = torch.empty(NUM_PARTS, dtype=torch.int64, device=dev)
    bounds = torch.empty(NUM_PARTS, 2, 3, device=dev)
    ppts, weights, tpose, parts = f(ppts), f(weights), f(tpose), parts.to(torch.int64).contiguous()      # held: pointers outlive the call
    _abi.check(_abi.lib().invr_pack_parts(_abi.ptr(ppts), _abi.ptr(weights), _abi.ptr(parts, torch.int64),
                                          _abi.ptr(tpose), V, W, V, float(bbox_overlap), _abi.ptr(part_pts), _abi.ptr(part_pbw),
                                          _abi.ptr(lengths2, torch.int64), _abi.ptr(bounds), _abi.stream_ptr()))
    M = int(lengths2.max())                                   # max_length trim (:591-593), one host sync as in NumPy
    return part_pts[:, :M].contiguous(), part_pbw[:, :M].contiguous(), lengths2, bounds
