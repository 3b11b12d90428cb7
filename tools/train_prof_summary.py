"""Markdown summary of one tools/kt_train.sh result directory (gpurun_out/<tag>): tools/train_bench.py output + the rocprofv3
--kernel-trace --stats table of the training iteration.
usage: python tools/train_prof_summary.py gpurun_out/r2i_train profiles/r2i_training_step.md "<title>" """
import csv
import sys

R, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
out = ['# ' + title + '\n']
out.append('Commands (MI355X box, `bash tools/kt_train.sh <tag>`): `python tools/train_bench.py --iters 40` (BASELINE configs[4] shape: one 32x32 patch = 1024 rays x\n'
           '128 samples, full-size inb_377 model, fused forward + backward + FusedAdam over all 286 M parameters, gradient arena), then the same\n'
           'under `rocprofv3 --kernel-trace --stats --output-format csv`.\n')
out.append('## tools/train_bench.py\n```\n' + ''.join(l for l in open(R + '/train_bench.log') if 'amdgpu.ids' not in l).strip() + '\n```\n')
rows = list(csv.DictReader(open(R + '/tr/tr_kernel_stats.csv')))
calls_adam = max(1, int(next(r['Calls'] for r in rows if r['Name'].startswith('k_adam('))))
short = lambda n: n.replace('void ', '').split('(')[0][:70]
out.append('## kernel-trace stats (%d iterations incl. warm-up and the synchronised split)\n| kernel | calls | calls / iteration | avg_us | max_us | us / iteration | pct |\n|---|---|---|---|---|---|---|' % calls_adam)
tot = 0.0
for r in rows[:34]:
    per = float(r['TotalDurationNs']) / 1e3 / calls_adam
    tot += per
    out.append('| %s | %s | %.1f | %.1f | %.1f | %.1f | %s |' % (short(r['Name']), r['Calls'], int(r['Calls']) / calls_adam, float(r['AverageNs']) / 1e3,
                                                           float(r['MaxNs']) / 1e3, per, r['Percentage']))
out.append('\nSum of the listed kernels: %.0f us of GPU time per iteration.' % tot)
open(dst, 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[-12:]))
