"""GPU: the deformer's UV-volume look-up with 24-bit index math (round 6: volumes of <= 2^24 elements, every scene of the suite) against
the 64-bit form the kernels keep for larger volumes (lib/utils/blend_utils.py:501-555 has no size limit).  The same frame is rendered
with its UV volume and with that volume ZERO-PADDED past 2^24 elements, the bounds extended by whole voxels so that every sample falls on
the same voxels with the same weights up to the rounding of the normalised coordinate."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_uv_volume_beyond_2_pow_24_elements_takes_the_64_bit_index_path_and_agrees():
    import invr  # noqa: F401
    from invr import scene, params
    from invr.config import make_cfg
    from invr.network import Network
    from invr.renderer import Renderer
    dev = 'cuda:0'
    cfg = make_cfg(table_log2=12, N_samples=32)
    sd = params.init_state_dict(cfg, seed=5)
    batch_np, _ = scene.make_scene(48, 48, seed=1)
    net = Network(cfg=cfg)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    rend = Renderer(net)
    small = {k: v.to(dev) for k, v in scene.to_torch(batch_np).items()}
    tuv = np.asarray(batch_np['tuv'])
    tb = np.asarray(batch_np['tbounds'], np.float64)
    batched = tuv.ndim == 5
    vol = tuv[0] if batched else tuv
    dx, dy, dz, c = vol.shape
    assert c == 2 and dx * dy * dz * c <= 1 << 24                      # the suite's scenes take the 24-bit path
    # pad x and y behind the volume until it holds more than 2^24 elements; voxel spacing unchanged
    fx = int(np.ceil(np.sqrt((1 << 24) * 1.05 / (dx * dy * dz * c)))) + 1
    nx, ny = dx * fx, dy * fx
    big = np.zeros((nx, ny, dz, c), np.float32)
    big[:dx, :dy] = vol
    assert big.size > 1 << 24
    b = tb.reshape(-1, 2, 3)[0] if tb.ndim == 3 else tb.reshape(2, 3)
    nb = b.copy()
    nb[1, 0] = b[0, 0] + (b[1, 0] - b[0, 0]) * (nx - 1) / (dx - 1)
    nb[1, 1] = b[0, 1] + (b[1, 1] - b[0, 1]) * (ny - 1) / (dy - 1)
    big_np = dict(batch_np)
    big_np['tuv'] = big[None] if batched else big
    big_np['tbounds'] = nb.astype(np.float32).reshape(np.asarray(batch_np['tbounds']).shape)
    bigb = {k: v.to(dev) for k, v in scene.to_torch(big_np).items()}
    with torch.no_grad():
        a = rend.render(small)
        stats_a = rend.last_stats.cpu().numpy().copy()
        r_a, raw_a = a['rgb_map'].cpu().clone(), a['raw'].cpu().clone()
        b_ = rend.render(bigb)
        stats_b = rend.last_stats.cpu().numpy().copy()
        r_b, raw_b = b_['rgb_map'].cpu().clone(), b_['raw'].cpu().clone()
    # geometry does not depend on the UV volume: identical survivors and pair lists
    assert (stats_a[:6] == stats_b[:6]).all(), (stats_a[:8], stats_b[:8])
    assert int(stats_a[0]) > 1000
    # the normalised coordinate of a point differs by rounding between the two bounds (~1e-7 relative), the residual deformer is
    # steep in it (nearest-vertex UV volume): nearly every ray-sample agrees to fp32 noise, none is off by an index's worth
    d = (raw_a - raw_b).abs().reshape(-1, 4).max(1)[0]
    assert float((d > 1e-4).float().mean()) < 2e-3, float((d > 1e-4).float().mean())
    assert float((r_a - r_b).abs().max()) < 2e-2
    assert float((r_a - r_b).abs().mean()) < 1e-4
