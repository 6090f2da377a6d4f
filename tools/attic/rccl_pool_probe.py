# RCCL next to the raised signal pool (the package import sets ROC_SIGNAL_POOL_SIZE=4096): a one-rank process group,
# all_reduce + barrier, then the batch harness (run_batch) on one tiny subject - what bench.py --gpus N does per rank.
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unires_amd  # noqa: F401
import torch.distributed as dist
from unires_amd import batch
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29517')
print('ROC_SIGNAL_POOL_SIZE', os.environ.get('ROC_SIGNAL_POOL_SIZE'))
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
t = torch.ones(1 << 20, device='cuda')
for _ in range(5):
    dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
print('all_reduce ok', float(t[0]))
import bench
x, y, z, w, rho, sett = bench.build_subject(bench.WORKLOADS['tiny_32c3_thick2'], torch.device('cuda:0'), seed=1)
tmp = torch.zeros_like(y[0].dat)


def rec(s):
    for it in range(5):
        unires_amd._update_admm(x, y, z, w, rho, tmp, None, it, sett)
    return float(y[0].dat.double().sum())


r = batch.run_batch(1, rec, device=torch.device('cuda:0'))
print('run_batch ok', r['subjects_per_sec'] > 0, r['results'])
dist.destroy_process_group()
print('DONE')
