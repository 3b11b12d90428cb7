"""Golden vectors for on-device ray generation, produced by the REFERENCE's own
get_rays_within_bounds (lib/utils/if_nerf/if_nerf_data_utils.py:313-327 -> :24-38, :92-107).
Run here only:  python tests/golden/make_golden_rays.py   (cv2 / trimesh are stubbed: unused by these functions)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as mg   # noqa: E402


def main():
    for name in ('cv2', 'trimesh'):
        sys.modules[name] = types.ModuleType(name)
    mg.import_reference(16)
    from lib.utils.if_nerf import if_nerf_data_utils as du
    import invr
    from invr import scene
    out = {}
    for tag, (H, W, cd) in {'a': (48, 40, 3.0), 'b': (33, 57, 1.6)}.items():
        b, ex = scene.make_scene(H, W, seed=0, cam_dist=cd)
        K, R, T = ex['K'], ex['Rc'], ex['Tc']
        wb = b['wbounds'][0]
        ray_o, ray_d, near, far, mask = du.get_rays_within_bounds(H, W, K, R, T, wb)
        out.update({tag + '_K': K, tag + '_R': R, tag + '_T': T, tag + '_bounds': wb, tag + '_HW': np.array([H, W]),
                    tag + '_ray_o': ray_o, tag + '_ray_d': ray_d, tag + '_near': near, tag + '_far': far, tag + '_mask': mask})
        # the scene generator's own restatement must agree bit for bit
        assert np.array_equal(ray_o, b['ray_o'][0]) and np.array_equal(ray_d, b['ray_d'][0])
        assert np.array_equal(near, b['near'][0]) and np.array_equal(far, b['far'][0])
        assert np.array_equal(mask.reshape(-1), b['mask_at_box'][0])
    path = os.path.join(HERE, 'rays_small.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    main()
