"""Golden vectors for the per-part KNN reference sets (row f4: invr_pack_parts).

Runs in the BUILD container only: the reference has no function for this step — it is inline code of
Dataset.__getitem__ (lib/datasets/h36m/tpose_dataset.py:569-591) — so the generator executes exactly those source lines of the
reference checkout (read at generation time, never copied into this repository) on the synthetic body of invr.scene and stores
inputs + outputs in tests/golden/parts_small.npz.
    python tests/golden/make_golden_parts.py
"""
import os
import sys
import textwrap
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = '/root/reference/lib/datasets/h36m/tpose_dataset.py'


def main():
    import invr  # noqa: F401
    from invr import scene
    src = open(REF).read().split('\n')
    start = next(i for i, l in enumerate(src) if l.strip() == "N, D = self.meta_smpl['weights'].shape")
    end = next(i for i in range(start, len(src)) if src[i].strip() == "part_pbw = part_pbw[:, :max_length, :]")
    code = textwrap.dedent('\n'.join(src[start:end + 1]))
    out = {}
    for tag, seed, overlap in (('a', 0, 0.2), ('b', 7, 0.05)):
        tverts, weights, parts, _ = scene.make_body(seed)
        rng = np.random.RandomState(seed + 5)
        poses = rng.uniform(-1, 1, (24, 3)) * 0.4
        poses[0] = 0
        A = scene.rigid_transformation(poses, scene._J, scene.PARENTS)
        big = np.zeros(72); big[5] = np.deg2rad(30); big[8] = np.deg2rad(-30)
        ppts = scene.lbs(tverts, weights, A)
        tpose = scene.lbs(tverts, weights, scene.rigid_transformation(big.reshape(24, 3), scene._J, scene.PARENTS))
        ns = {'np': np, 'NUM_PARTS': 5, 'ppts': ppts, 'tpose': tpose, 'ret': {},
              'self': types.SimpleNamespace(meta_smpl={'weights': weights, 'parts': parts}),
              'cfg': types.SimpleNamespace(bbox_overlap=overlap)}
        exec(code, ns)
        out.update({tag + '_ppts': ppts, tag + '_weights': weights, tag + '_parts': parts.astype(np.int64), tag + '_tpose': tpose,
                    tag + '_overlap': np.float32(overlap), tag + '_part_pts': ns['part_pts'], tag + '_part_pbw': ns['part_pbw'],
                    tag + '_lengths2': ns['lengths2'].astype(np.int64), tag + '_bounds': ns['bounds']})
    # keep the fixture small: the inputs are regenerated from the seeds by the tests, only the outputs' digests + a few rows are stored
    small = {}
    for k, v in out.items():
        if k.endswith(('_ppts', '_weights', '_parts', '_tpose')):
            continue
        small[k] = v if v.size <= 64 else np.array([np.float64(v.astype(np.float64).sum()), np.float64(np.abs(v.astype(np.float64)).sum()),
                                                     np.float64((v.astype(np.float64) * (np.arange(v.size).reshape(v.shape) % 97)).sum())])
        if v.size > 64:
            small[k + '_shape'] = np.array(v.shape)
            small[k + '_rows'] = v.reshape(-1, v.shape[-1])[:: max(1, v.reshape(-1, v.shape[-1]).shape[0] // 50)][:50].copy()
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'parts_small.npz'), **small)
    print('wrote parts_small.npz', {k: getattr(v, 'shape', None) for k, v in small.items()})


if __name__ == '__main__':
    main()
