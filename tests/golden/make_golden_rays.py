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


def main_f4():
    """Row f4: get_rigid_transformation (if_nerf_data_utils.py:545-577, batch_rodrigues :523-542) goldens."""
    for name in ('cv2', 'trimesh'):
        sys.modules.setdefault(name, types.ModuleType(name))
    if 'lib.config' not in sys.modules:
        mg.import_reference(16)
    from lib.utils.if_nerf import if_nerf_data_utils as du
    from invr import scene
    out = {}
    rng = np.random.RandomState(4)
    for tag in ('p', 'q', 'z'):
        poses = rng.uniform(-1.2, 1.2, (24, 3)) * (0 if tag == 'z' else 1)
        joints = (scene._J + rng.uniform(-0.02, 0.02, (24, 3))).astype(np.float64)
        A = du.get_rigid_transformation(poses, joints, scene.PARENTS)
        out.update({tag + '_poses': poses, tag + '_joints': joints, tag + '_A': A})
    out['parents'] = scene.PARENTS.astype(np.int32)
    path = os.path.join(HERE, 'rigid_small.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    main()
    main_f4()
