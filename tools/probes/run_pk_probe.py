"""Runs tools/probes/pk_probe.hip beside the frame pipeline: K frames in flight (one hipGraph replay after the other) on one stream, the
probe's small packed-vs-scalar checking waves on another, everywhere on the chip.  Prints the mismatching lanes per instruction form and
16-lane quarter of the wave.   python tools/probes/run_pk_probe.py [replays] [alone]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                          # noqa: E402
import invr                           # noqa: E402,F401
from invr import frames as iframes    # noqa: E402
from invr.config import make_cfg      # noqa: E402
import bench                          # noqa: E402

KINDS = ['v_pk_mul_f32', 'v_pk_add_f32', 'v_pk_fma_f32', 'v_pk_mul_f32 op_sel_hi:[0,1]', 'v_pk_fma_f32 op_sel:[1,0,0]', 'dependent chain pk_mul -> pk_fma -> pk_add',
         'v_mul_f32 ; v_pk_mul_f32 ; v_add_f32 of both', 'v_pk_mov_b32 op_sel:[1,0]']


def main():
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    alone = len(sys.argv) > 2 and sys.argv[2] == 'alone'
    dev = torch.device('cuda', 0)
    P = C.CDLL(os.path.join(ROOT, 'variants', 'libpkprobe.so'))
    counts = torch.zeros(40, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(dev)
    fs = None
    if not alone:
        cfg = make_cfg(N_samples=128)
        cfg['eval_row_sums'] = True
        net = bench.build_model(cfg, dev)
        _, batches = bench.frame_batches(512, 1.8, 10, dev)
        fns, n_rays, keep = iframes.shard_render_fns(net, batches, 128, 0, 1, want_raw=True)
        fs = iframes.FrameSet(fns, n_rays, device=dev)
    torch.cuda.synchronize()
    for rep in range(replays):
        if fs is not None:
            fs.replay()
        with torch.cuda.stream(side):
            # 2048 single-wave workgroups x 20000 iterations: ~ the length of a replay, a wave or two on every SIMD beside the frames' waves
            assert P.pk_probe_launch(C.c_void_p(side.cuda_stream), 2048, 20000, C.c_void_p(counts.data_ptr()), C.c_float(1.0 + 0.01 * rep)) == 0
        torch.cuda.synchronize()
    c = counts.cpu().tolist()
    out = {'replays': replays, 'frames_beside': not alone,
           'mismatching_lane_checks': {KINDS[k]: {'quarters_0_15_16_31_32_47_48_63': c[k * 4:k * 4 + 4], 'checks': c[32 + k]} for k in range(8)}}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
