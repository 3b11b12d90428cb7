"""Experiment: can a world-size-1 RCCL all_gather_into_tensor be captured in a hipGraph here?"""
import os, sys, faulthandler
faulthandler.enable()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
import torch, torch.distributed as dist
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
mode = sys.argv[1]
send = torch.arange(4096 * 4, device=dev, dtype=torch.float32).view(4096, 4)
recv = torch.empty(4096, 4, device=dev)
src = torch.randperm(4096, device=dev)
dist.all_gather_into_tensor(recv, send); torch.cuda.synchronize()
print('eager ok', bool(torch.equal(recv, send)), flush=True)
g = torch.cuda.CUDAGraph()
if mode == 'plain':
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        dist.all_gather_into_tensor(recv, send)
        full = recv.index_select(0, src)
elif mode == 'global':
    with torch.cuda.graph(g):
        dist.all_gather_into_tensor(recv, send)
        full = recv.index_select(0, src)
elif mode == 'async':
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        w = dist.all_gather_into_tensor(recv, send, async_op=True)
        w.wait()
        full = recv.index_select(0, src)
print('captured', flush=True)
recv.zero_()
for _ in range(3): g.replay()
torch.cuda.synchronize()
print(mode, 'replay ok', bool(torch.equal(full, send[src])), flush=True)
dist.destroy_process_group()
