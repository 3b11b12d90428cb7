"""TEST INFRASTRUCTURE — builds tests/hostsim/_build/libinvr_hostsim.so: the kernel sources of instant-nvr_amd/csrc compiled for the
HOST against tests/hostsim/hip/hip_runtime.h (a wave machine on fibers, see README.md) so that CPU-only tests can run the very
kernel code of the product at toy sizes.  The sources are used as they are, except for three device-only constructs that have no
host spelling and are rewritten on the fly (the csrc tree stays free of host-simulation conditionals):
  * `extern __shared__ T name[];`                 -> a pointer to the simulated dynamic LDS
  * the v_min_f64 / v_max_f64 inline assembly      -> fmin / fmax (the keys are finite non-negative doubles, k_knn.hip:68-72)
  * the empty `asm volatile("" : "+v"...)` fences  -> dropped (also `amdgpu_waves_per_eu` occupancy attributes)
Nothing under instant-nvr_amd/ imports this module or loads its output."""
import concurrent.futures as cf
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'instant-nvr_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
LIB = os.path.join(OUT, 'libinvr_hostsim.so')
CXX = os.environ.get('HOSTSIM_CXX') or '/opt/rocm/lib/llvm/bin/clang++'
def _cpu_has(flag):
    try:
        return any(flag in line.split() for line in open('/proc/cpuinfo') if line.startswith('flags'))
    except OSError:
        return False


# -ffp-contract=off as the product build (FMAs only where the source says fmaf()); -mfma turns fmaf() into the instruction where the host
# has it (libm's fmaf is exact too, only slower)
FLAGS = ['-x', 'c++', '-std=c++17', '-O1', '-g0', '-fPIC', '-fno-strict-aliasing', '-ffp-contract=off'] + (['-mfma'] if _cpu_has('fma') else []) + [
         '-Wno-unused-function', '-Wno-unused-value', '-Wno-unknown-pragmas', '-Wno-pass-failed',
         '-I', HERE, '-I', CSRC]

REWRITES = [
    (re.compile(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];'),
     r'\1* \2 = reinterpret_cast<\1*>(hostsim::dyn_lds());'),
    (re.compile(r'asm\("v_min_f64 %0, %1, %2" : "=v"\((\w+)\) : "v"\((\w+)\), "v"\((\w+)\)\);'), r'\1 = fmin(\2, \3);'),
    (re.compile(r'asm\("v_max_f64 %0, %1, %2" : "=v"\((\w+)\) : "v"\((\w+)\), "v"\((\w+)\)\);'), r'\1 = fmax(\2, \3);'),
    (re.compile(r'asm\("v_min_f32 %0, %1, %2" : "=v"\((\w+)\) : "v"\(([^)]+)\), "v"\(([^)]+)\)\);'), r'\1 = fminf(\2, \3);'),
    (re.compile(r'asm volatile\("" : [^;]*\);'), r''),
    (re.compile(r'__attribute__\(\(amdgpu_waves_per_eu\([^)]*\)\)\)'), r''),        # (an occupancy request of the device compiler)
]


def strip_experiments(text):
    """Drop the device-only experiment regions of a source (`#if WARP_EXP ...` with its `#else` branch kept, `#ifdef WARP_VERIFY`,
    `#if defined(WARP_VERIFY) ...`, `#ifdef MLPB_PROF`): debug kernels of investigation builds that read hardware registers; the host build is the product's."""
    out, stack = [], []          # stack entries: [is_experiment, keep_now]
    for line in text.split('\n'):
        t = line.strip()
        if t.startswith('#if'):
            exp = ('WARP_EXP' in t or 'WARP_VERIFY' in t or 'MLPB_PROF' in t) and not t.startswith('#ifndef')
            stack.append([exp, not exp])
            if exp:
                continue
        elif t.startswith('#else') and stack and stack[-1][0]:
            stack[-1][1] = True
            continue
        elif t.startswith('#endif') and stack:
            exp, _ = stack.pop()
            if exp:
                continue
        if all(keep for _, keep in stack):
            out.append(line)
    return '\n'.join(out)


# sanitizer builds only: a poisoned 256-byte red zone behind every array carved out of a caller's workspace (invr_abi.hip: Carver), so
# that a kernel running over the end of ITS array trips ASan instead of landing in the neighbouring array of the same allocation
ASAN_REWRITES = [
    (re.compile(r'(struct Carver \{.*?off \+= n \* sizeof\(T\);)', re.S),
     r'\1\n        { const size_t rz = off; off = align_up(off, 256) + 256; if ((uintptr_t)base > (1u << 20) && (uintptr_t)base != (uintptr_t(1) << 40)) __asan_poison_memory_region(base + rz, off - rz); }'),
    (re.compile(r'(\(3 \+ 3 \+ EMB_K\) \* sizeof\(float\) \+ 3 \* 256), 256\);'), r'\1 + 4096, 256);'),      # invr_part_field_workspace: room for them
    (re.compile(r'(// ---- workspace carve -+)'), r'\1\nextern "C" void __asan_poison_memory_region(void const volatile*, size_t);'),
]


# -DHOSTSIM_KNN_PERM (tools/knn_order_model.py): k_knn_pairs reads its survivors through a permutation set from outside, to count the
# sweep's work (the -DKNN_PROF counters) under other survivor orders than the compaction's.  Results of such a run are garbage.
def knn_perm_rewrite(text, name):
    if name != 'k_knn.hip':
        return text
    text = text.replace('w.active_idx[slot]', 'w.active_idx[hostsim_knn_perm ? hostsim_knn_perm[slot] : slot]')
    return text.replace('#define KNN_BLOCK 256', 'static const int* hostsim_knn_perm = nullptr;\n'
                        'extern "C" void hostsim_set_knn_perm(const int* p) { hostsim_knn_perm = p; }\n#define KNN_BLOCK 256', 1)


def transform(text, name, asan=False, knn_perm=False):
    if knn_perm:
        text = knn_perm_rewrite(text, name)
    text = strip_experiments(text)
    for rx, rep in REWRITES + (ASAN_REWRITES if asan else []):
        text = rx.sub(rep, text)
    code = re.sub(r'//[^\n]*', '', text)
    if re.search(r'\basm\b', code) or 'extern __shared__' in code:
        raise RuntimeError('hostsim/build.py: %s holds a device-only construct without a rewrite rule' % name)
    return text


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def digest():
    h = hashlib.sha256()
    for d, names in ((CSRC, sorted(os.listdir(CSRC))), (HERE, ['build.py', 'hostsim_rt.cpp', os.path.join('hip', 'hip_runtime.h')]),
                     (os.path.join(ROOT, 'include'), ['invr.h'])):
        for n in names:
            p = os.path.join(d, n)
            if os.path.isfile(p) and not n.endswith(('.so', '.o')):
                h.update(n.encode())
                h.update(open(p, 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _compile(job):
    src, obj, extra = job
    r = subprocess.run([CXX] + FLAGS + extra + ['-c', src, '-o', obj], capture_output=True, text=True)
    return src, r.returncode, r.stderr


def build(force=False, verbose=False, extra=()):
    """-> path of the library (built when the sources changed).  Every set of extra flags (sanitizer / experiment builds) has its own
    directory, so that a test process never finds the library it has loaded replaced by another build."""
    OUT = os.path.join(globals()['OUT'], hashlib.sha256(' '.join(extra).encode()).hexdigest()[:8]) if extra else globals()['OUT']
    LIB = os.path.join(OUT, 'libinvr_hostsim.so')
    os.makedirs(OUT, exist_ok=True)
    stamp = os.path.join(OUT, 'digest.txt')
    dg = digest() + ' ' + ' '.join(extra)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dg:
        return LIB
    jobs = []
    for f in sources():
        cpp = os.path.join(OUT, f[:-4] + '.cpp')
        text = transform(open(os.path.join(CSRC, f)).read(), f, asan='-fsanitize=address' in extra, knn_perm='-DHOSTSIM_KNN_PERM' in extra)
        with open(cpp, 'w') as fh:
            fh.write('#line 1 "%s"\n' % os.path.join(CSRC, f))
            fh.write(text)
        jobs.append((cpp, cpp[:-4] + '.o', list(extra)))
    jobs.append((os.path.join(HERE, 'hostsim_rt.cpp'), os.path.join(OUT, 'hostsim_rt.o'), list(extra)))
    failed = False
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for src, rc, err in ex.map(_compile, jobs):
            if rc != 0 or (verbose and err):
                sys.stderr.write('---- %s\n%s\n' % (src, err[-6000:]))
            failed |= rc != 0
    if failed:
        raise RuntimeError('hostsim build failed')
    r = subprocess.run([CXX, '-shared', '-o', LIB] + list(extra) + [j[1] for j in jobs] + ['-lm'], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hostsim link failed:\n' + r.stderr[-4000:])
    with open(stamp, 'w') as fh:
        fh.write(dg)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
