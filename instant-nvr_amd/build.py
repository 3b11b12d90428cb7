"""Build libinvr.so (HIP, gfx950) in-tree with hipcc.  `python -m invr.build` or build.build()."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libinvr.so')
SOURCES = ['invr_abi.hip', 'k_cull.hip', 'k_knn.hip', 'k_warp.hip', 'k_encode.hip', 'k_mlp.hip', 'k_composite.hip', 'k_rays.hip', 'k_prep.hip', 'k_optim.hip', 'k_mlp_bwd.hip', 'k_train.hip']
HEADERS = ['common.h', 'pipeline.h', 'grid_generic.h', 'mlp_common.h', 'train.h', os.path.join('..', '..', 'include', 'invr.h')]
# -ffp-contract=off: FMAs only where the source says fmaf(), so the discrete decisions of the path
# (cull / flag thresholds, integer cell selection) see the same fp32 arithmetic as the reference.
# -fno-slp-vectorize (round 6): no COMPILER-generated packed-fp32 math (v_pk_mul / v_pk_add / v_pk_fma_f32 with op_sel shuffles).  The SLP
# vectorizer had packed 82 of k_warp_pairs' multiplies and adds; with frames in flight (waves of other kernels — MFMA kernels — on the same
# SIMD) that kernel then returned, in ~1 of 300 frames, wrong values in lanes 48..63 of single waves, always in results of such packed
# sequences (every loaded and blended input of the failing pairs verified correct: profiles/r6_replay_mismatch.md).  Without them: 0
# mismatches in 10,000 frames, same frame time.  The hand-written packed arithmetic of k_knn.hip stays (its workgroups own their SIMDs).
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize', '-Wall', '-Wno-unused-function']


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    cc = hipcc()
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    stamp = os.path.join(objdir, 'flags.txt')              # (objects built with other flags are stale too)
    if not os.path.exists(stamp) or open(stamp).read() != ' '.join(FLAGS):
        force = True
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace('.hip', '.o'))
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [cc] + FLAGS + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    with ThreadPoolExecutor(max_workers=max(1, min(8, len(jobs)))) as ex:
        for (src, obj), r in ex.map(compile_one, jobs):
            if verbose and (r.stdout.strip() or r.stderr.strip()):
                sys.stderr.write(r.stdout + r.stderr)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed on %s' % src)
    with open(stamp, 'w') as f:
        f.write(' '.join(FLAGS))
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError('link failed')
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
