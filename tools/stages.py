import sys, os, json, subprocess
# run bench (no cpu baseline) and print stage times compactly; extra args are passed through
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tools/)
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    st = d['stage_ms_per_step']
    print('%s ms=%.3f val=%.3e cull=%.2f knn=%.2f warp=%.2f enc=%.2f mlp=%.2f comp=%.2f' % (os.environ.get('TAG', ''), d['ms_per_step'], d['value'],
          st['cull'], st['knn'], st['warp'], sum(st['encode_%d' % p] for p in range(5)), sum(st['mlp_%d' % p] for p in range(5)), st['composite']))
except Exception as e:
    print('FAILED', e, out.stdout[-500:], out.stderr[-2000:])
