"""The N>1 path on CPU: world_size-2 gloo processes run the batch partitioner
(no GPU, no compute kernels - SURVEY 8(e): replicas only, scalars-only collectives)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_subjects, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from unires_amd import batch
    r, w, _ = batch.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)

    def reconstruct(s):  # stand-in for a subject's reconstruction: deterministic scalar
        g = torch.Generator().manual_seed(1000 + s)
        return torch.rand(64, generator=g).double().sum().item()

    res = batch.run_batch(n_subjects, reconstruct)
    torch.save(res, os.path.join(out_dir, 'rank%d.pt' % rank))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_partition_covers_every_subject_once():
    from unires_amd.batch import partition
    for n in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            got = sorted(s for r in range(world) for s in partition(n, world, r))
            assert got == list(range(n))
            sizes = [len(partition(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        partition(4, 2, 2)


@pytest.mark.parametrize('n_subjects', [5, 8])
def test_two_rank_gloo_batch(tmp_path, n_subjects):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_subjects, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r), weights_only=False)
           for r in range(world)]
    expect = {}
    for s in range(n_subjects):
        g = torch.Generator().manual_seed(1000 + s)
        expect[s] = torch.rand(64, generator=g).double().sum().item()
    for r in range(world):
        assert res[r]['world_size'] == 2 and res[r]['rank'] == r
        assert res[r]['mine'] == list(range(r, n_subjects, 2))
        assert res[r]['results'] == expect                      # every rank sees all results
        assert res[r]['elapsed'] == res[0]['elapsed']            # max over ranks, agreed
        assert res[r]['subjects_per_sec'] == pytest.approx(n_subjects / res[0]['elapsed'])


def test_single_process_needs_no_process_group():
    from unires_amd import batch
    out = batch.run_batch(3, lambda s: s * 2.0)
    assert out['results'] == {0: 0.0, 1: 2.0, 2: 4.0} and out['world_size'] == 1


def test_bench_self_spawn_launches_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus N` without a launcher re-launches itself through
    torch.distributed.run (bench.spawn_ranks).  The command line it builds is run here for real -
    two gloo ranks on CPU - with a probe script in place of bench.py."""
    import subprocess
    import sys
    import bench
    cmd = bench.spawn_ranks(2, ['--gpus', '2', '--steps', '3'], dry_run=True)
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '2'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ['--gpus', '2', '--steps', '3']
    probe = tmp_path / 'probe.py'
    probe.write_text(
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import torch, torch.distributed as dist\n"
        "from unires_amd import batch\n"
        "rank, world, local = batch.init_from_env(backend='gloo')\n"
        "t = torch.tensor([float(rank + 1)], dtype=torch.float64)\n"
        "dist.all_reduce(t, op=dist.ReduceOp.MAX)\n"
        "dist.barrier()\n"
        "if rank == 0:\n"
        "    print('n_gpus=%%d max=%%g args=%%s' %% (world, t.item(), ' '.join(sys.argv[1:])))\n"
        "dist.destroy_process_group()\n" % os.path.dirname(os.path.abspath(bench.__file__)))
    cmd[i] = str(probe)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'n_gpus=2 max=2 args=--gpus 2 --steps 3' in out.stdout
