"""Seeded synthetic stand-in for one ZJU-MoCap / MonoCap sample.

The licensed datasets are not available (reference docs/install.md:24-53), so
every BASELINE config runs on this generator, which emits the exact ``batch``
dict contract of the reference's dataset (lib/datasets/h36m/tpose_dataset.py:454-600,
SURVEY.md §8b): a capsule "stick figure" with 24 SMPL-like joints and 6890
vertices, soft skinning weights, LBS matrices ``A``/``big_A``
(if_nerf_data_utils.py:545-577 semantics), per-part KNN reference sets, 0.025 m
blend-weight/UV volumes (tools/prepare_zjumocap.py:152-165 semantics), a pinhole
camera and ray/near/far generation (if_nerf_data_utils.py:24-38, 92-107).

NumPy only; deterministic for a given seed.
"""
import numpy as np

from .config import NUM_PARTS, PART_NAMES, PART_BW_MAP

N_VERTS = 6890
PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21])

# rest joints, metres, y up, pelvis at y=-0.24 so that the inb_377 part boxes fit
_J = np.array([
    [0.00, 0.00, 0.00], [0.07, -0.09, 0.00], [-0.07, -0.09, 0.00], [0.00, 0.11, 0.00],
    [0.10, -0.47, 0.00], [-0.10, -0.47, 0.00], [0.00, 0.25, 0.00], [0.09, -0.87, -0.03],
    [-0.09, -0.87, -0.03], [0.00, 0.30, 0.00], [0.11, -0.93, 0.09], [-0.11, -0.93, 0.09],
    [0.00, 0.51, 0.00], [0.08, 0.42, 0.00], [-0.08, 0.42, 0.00], [0.00, 0.60, 0.02],
    [0.18, 0.44, 0.00], [-0.18, 0.44, 0.00], [0.44, 0.44, 0.00], [-0.44, 0.44, 0.00],
    [0.69, 0.44, 0.00], [-0.69, 0.44, 0.00], [0.78, 0.44, 0.00], [-0.78, 0.44, 0.00]], dtype=np.float64)
_J[:, 1] -= 0.24
# capsule radius of the bone that ENDS at joint j (parent(j) -> j)
_RAD = np.array([0, .09, .09, .12, .07, .07, .12, .05, .05, .12, .04, .04, .06, .07, .07, .09,
                 .06, .06, .045, .045, .04, .04, .035, .035])


def rodrigues(rvec):
    """axis-angle (...,3) -> rotation matrices (...,3,3)."""
    rvec = np.asarray(rvec, dtype=np.float64)
    ang = np.linalg.norm(rvec, axis=-1, keepdims=True)
    axis = rvec / np.maximum(ang, 1e-12)
    x, y, z = axis[..., 0], axis[..., 1], axis[..., 2]
    zero = np.zeros_like(x)
    K = np.stack([zero, -z, y, z, zero, -x, -y, x, zero], -1).reshape(rvec.shape[:-1] + (3, 3))
    s, c = np.sin(ang)[..., None], np.cos(ang)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def rigid_transformation(poses, joints, parents):
    """Per-joint 4x4 LBS transforms (restates if_nerf_data_utils.py:545-577)."""
    rot = rodrigues(poses.reshape(-1, 3))
    rel = joints.copy()
    rel[1:] -= joints[parents[1:]]
    local = np.zeros((24, 4, 4))
    local[:, :3, :3] = rot
    local[:, :3, 3] = rel
    local[:, 3, 3] = 1
    chain = [local[0]]
    for i in range(1, 24):
        chain.append(chain[parents[i]] @ local[i])
    T = np.stack(chain)
    jh = np.concatenate([joints, np.zeros((24, 1))], 1)
    T[..., 3] = T[..., 3] - np.einsum('jab,jb->ja', T, jh)
    return T.astype(np.float32)


def _seg_dist(p, a, b):
    ab = b - a
    t = np.clip(((p - a) @ ab) / max(float(ab @ ab), 1e-12), 0, 1)
    return np.linalg.norm(p - (a + t[:, None] * ab), axis=1)


def make_body(seed=0):
    """Rest-pose vertices (6890,3), skinning weights (6890,24), part id (6890,), uv (6890,2)."""
    rng = np.random.RandomState(seed)
    bones = [(PARENTS[j], j) for j in range(1, 24)]
    area = np.array([np.linalg.norm(_J[c] - _J[p]) * _RAD[c] + 2 * _RAD[c] ** 2 for p, c in bones])
    cnt = np.floor(area / area.sum() * N_VERTS).astype(int)
    cnt[np.argmax(cnt)] += N_VERTS - cnt.sum()
    verts = []
    for (p, c), n in zip(bones, cnt):
        a, b, r = _J[p], _J[c], _RAD[c]
        ax = (b - a) / np.linalg.norm(b - a)
        tmp = np.array([1., 0, 0]) if abs(ax[0]) < 0.9 else np.array([0, 1., 0])
        u = np.cross(ax, tmp); u /= np.linalg.norm(u)
        v = np.cross(ax, u)
        t = rng.uniform(-0.15, 1.15, n)            # overshoot -> rounded caps
        th = rng.uniform(0, 2 * np.pi, n)
        tc = np.clip(t, 0, 1)
        over = (t - tc) * np.linalg.norm(b - a)
        rr = np.sqrt(np.maximum(r * r - over * over, (0.2 * r) ** 2))
        pts = a + np.outer(tc, b - a) + np.outer(over, ax) + rr[:, None] * (np.outer(np.cos(th), u) + np.outer(np.sin(th), v))
        verts.append(pts)
    verts = np.concatenate(verts)
    # skinning weight of joint j ~ distance to the segments j -> child(j)
    d = np.full((N_VERTS, 24), 1e9)
    for j in range(24):
        kids = [c for c in range(24) if PARENTS[c] == j]
        if not kids:
            d[:, j] = np.linalg.norm(verts - _J[j], axis=1) + 0.03
        for c in kids:
            d[:, j] = np.minimum(d[:, j], _seg_dist(verts, _J[j], _J[c]))
    w = np.exp(-(d - d.min(1, keepdims=True)) / 0.05)
    w /= w.sum(1, keepdims=True)
    joint2part = np.zeros(24, dtype=np.int64)
    for pid, name in enumerate(PART_NAMES):
        joint2part[PART_BW_MAP[name]] = pid
    parts = joint2part[np.argmax(w, 1)]
    # a smooth-ish surface parameterisation in [0,1]^2
    uv = np.stack([np.arctan2(verts[:, 2], verts[:, 0]) / (2 * np.pi) + 0.5,
                   (verts[:, 1] - verts[:, 1].min()) / np.ptp(verts[:, 1])], 1)
    return verts.astype(np.float32), w.astype(np.float32), parts, uv.astype(np.float32)


def lbs(verts, weights, A):
    Aw = np.einsum('vj,jab->vab', weights.astype(np.float64), A.astype(np.float64))
    return (np.einsum('vab,vb->va', Aw[:, :3, :3], verts.astype(np.float64)) + Aw[:, :3, 3]).astype(np.float32)


def _volume(bounds, voxel=0.025):
    lo, hi = bounds
    dims = np.ceil((hi - lo) / voxel).astype(int) + 1
    axes = [lo[a] + voxel * np.arange(dims[a], dtype=np.float32) for a in range(3)]
    g = np.stack(np.meshgrid(*axes, indexing='ij'), -1).astype(np.float32)
    # the reference volumes span exactly [lo, lo+voxel*(dims-1)]; bounds are re-derived from that
    hi2 = np.array([axes[a][-1] for a in range(3)], dtype=np.float32)
    return g, np.stack([lo.astype(np.float32), hi2])


def get_rays(H, W, K, R, T):
    """Pixel rays (restates if_nerf_data_utils.py:24-38)."""
    o = -(R.T @ T).ravel()
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    xy1 = np.stack([i, j, np.ones_like(i)], 2)
    pc = xy1 @ np.linalg.inv(K).T
    pw = (pc - T.ravel()) @ R
    d = pw - o[None, None]
    d = d / np.linalg.norm(d, axis=2, keepdims=True)
    return np.broadcast_to(o, d.shape), d


def get_near_far(bounds, ray_o, ray_d):
    """Ray/AABB slab test (restates if_nerf_data_utils.py:92-107)."""
    norm_d = np.linalg.norm(ray_d, axis=-1, keepdims=True)
    vd = ray_d / norm_d
    vd[(vd < 1e-5) & (vd > -1e-10)] = 1e-5
    vd[(vd > -1e-5) & (vd < 1e-10)] = -1e-5
    tmin = (bounds[:1] - ray_o[:1]) / vd
    tmax = (bounds[1:2] - ray_o[:1]) / vd
    near = np.max(np.minimum(tmin, tmax), -1)
    far = np.min(np.maximum(tmin, tmax), -1)
    m = near < far
    return near[m] / norm_d[m, 0], far[m] / norm_d[m, 0], m


def make_scene(H=64, W=64, seed=0, frame=3, num_train_frame=100, bbox_overlap=0.2, pose_scale=0.5,
               cam_dist=3.0, crop=None, pose_seed=None):
    """Build one collated ``batch`` (numpy arrays with the leading batch dim of 1).

    ``crop`` = (y0, x0, h, w) keeps only the rays of that pixel window (training patches).
    ``pose_seed`` draws the pose / body orientation of the SAME body (``seed``) from another stream: the frames of a sequence
    (None = the pose of ``seed``, the frame every fixture and the round-1..3 bench lines use).
    Returns (batch, extras) where extras holds un-batched helper arrays (rest verts, ...).
    """
    from scipy.spatial import cKDTree
    rng = np.random.RandomState((seed if pose_seed is None else 7919 * pose_seed + seed) + 1000)
    tverts, weights, parts, vuv = make_body(seed)

    poses = rng.uniform(-1, 1, (24, 3)) * pose_scale / np.sqrt(3)
    poses[0] = 0
    A = rigid_transformation(poses, _J, PARENTS)
    big = np.zeros(72)
    big[5] = np.deg2rad(30); big[8] = np.deg2rad(-30)          # tpose_dataset.py:278-283
    big_A = rigid_transformation(big.reshape(24, 3), _J, PARENTS)

    ppts = lbs(tverts, weights, A)                              # posed SMPL ("pose space")
    tpose = lbs(tverts, weights, big_A)                         # canonical big pose
    Rh = rng.uniform(-0.3, 0.3, 3)
    Rw = rodrigues(Rh).astype(np.float32)
    Th = np.array([[0.05, 0.02, -0.03]], dtype=np.float32)
    wpts = (ppts @ Rw.T + Th).astype(np.float32)

    def aabb(x, pad):
        return np.stack([x.min(0) - pad, x.max(0) + pad]).astype(np.float32)

    # blend-weight volume in pose space: 24 weights + distance to the surface (here: nearest vertex)
    pg, pbounds = _volume(aabb(ppts, 0.05))
    dist, idx = cKDTree(ppts.astype(np.float64)).query(pg.reshape(-1, 3).astype(np.float64))
    pbw = np.concatenate([weights[idx], dist[:, None].astype(np.float32)], 1).reshape(pg.shape[:3] + (25,))
    # UV volume in canonical space
    tg, tbounds = _volume(aabb(tpose, 0.05))
    _, tidx = cKDTree(tpose.astype(np.float64)).query(tg.reshape(-1, 3).astype(np.float64))
    tuv = vuv[tidx].reshape(tg.shape[:3] + (2,))
    wbounds = aabb(wpts, 0.05)

    # per-part KNN reference sets (tpose_dataset.py:570-600)
    part_pts = np.zeros((NUM_PARTS, N_VERTS, 3), np.float32)
    part_pbw = np.zeros((NUM_PARTS, N_VERTS, 24), np.float32)
    lengths2 = np.zeros(NUM_PARTS, np.int64)
    bounds = np.zeros((NUM_PARTS, 2, 3), np.float32)
    for pid in range(NUM_PARTS):
        f = parts == pid
        lengths2[pid] = f.sum()
        part_pts[pid, :lengths2[pid]] = ppts[f]
        part_pbw[pid, :lengths2[pid]] = weights[f]
        bounds[pid, 0] = tpose[f].min(0) - bbox_overlap
        bounds[pid, 1] = tpose[f].max(0) + bbox_overlap
    M = int(lengths2.max())
    part_pts, part_pbw = part_pts[:, :M], part_pbw[:, :M]

    # camera: pinhole, f = 555 px at 512x512, looking at the body centre from +z
    f = 555.0 * H / 512.0
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], dtype=np.float64)
    centre = wpts.mean(0).astype(np.float64)
    Rc = rodrigues(np.array([np.pi, 0, 0])) @ rodrigues(np.array([0, 0.35, 0]))   # y down, look along -z
    Tc = (-Rc @ (centre + Rc.T @ np.array([0, 0, -cam_dist]))).reshape(3, 1)
    ray_o, ray_d = get_rays(H, W, K, Rc, Tc)
    ray_o = ray_o.reshape(-1, 3).astype(np.float32)
    ray_d = ray_d.reshape(-1, 3).astype(np.float32)
    near, far, mask_at_box = get_near_far(wbounds, ray_o, ray_d)
    if crop is not None:
        y0, x0, h, w = crop
        win = np.zeros((H, W), bool); win[y0:y0 + h, x0:x0 + w] = True
        keep = win.reshape(-1)[mask_at_box]
        near, far = near[keep], far[keep]
        mask_at_box = mask_at_box & win.reshape(-1)
    ray_o, ray_d = ray_o[mask_at_box], ray_d[mask_at_box]
    near, far = near.astype(np.float32), far.astype(np.float32)
    # synthetic supervision: a smooth colour field + a silhouette-ish occupancy
    pix = np.argwhere(mask_at_box.reshape(H, W))
    rgb = np.stack([0.5 + 0.5 * np.sin(pix[:, 0] * 0.11), 0.5 + 0.5 * np.cos(pix[:, 1] * 0.07),
                    0.5 + 0.5 * np.sin((pix[:, 0] + pix[:, 1]) * 0.05)], 1).astype(np.float32)
    mid = ray_o + ray_d * (0.5 * (near + far))[:, None]
    occupancy = (cKDTree(wpts.astype(np.float64)).query(mid.astype(np.float64))[0] < 0.12)

    latent_index = int(frame)
    b = {
        'ray_o': ray_o, 'ray_d': ray_d, 'near': near, 'far': far, 'rgb': rgb,
        'occupancy': occupancy.astype(np.uint8), 'mask_at_box': mask_at_box,
        'A': A, 'big_A': big_A, 'pbw': pbw.astype(np.float32), 'tuv': tuv.astype(np.float32),
        'pbounds': pbounds, 'tbounds': tbounds, 'wbounds': wbounds,
        'R': Rw, 'Th': Th, 'H': np.int64(H), 'W': np.int64(W),
        'frame_dim': np.float32(latent_index / num_train_frame),
        'latent_index': np.int64(latent_index),
        'ppts': ppts, 'part_pts': part_pts, 'part_pbw': part_pbw, 'lengths2': lengths2,
        'bounds': bounds,
    }
    batch = {k: np.asarray(v)[None] for k, v in b.items()}      # default_collate with batch_size 1
    extras = {'tverts': tverts, 'weights': weights, 'parts': parts, 'tpose': tpose, 'wpts': wpts,
              'K': K, 'Rc': Rc, 'Tc': Tc, 'poses': poses}
    return batch, extras


def to_torch(batch, device='cpu'):
    import torch
    out = {}
    for k, v in batch.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        out[k] = t.to(device)
    return out
