"""N > 1 path on CPU: two gloo processes shard an image list round-robin ("replicas only", no data-path collective) and the
per-rank statistics are gathered on the host and combined like bench.py does (pixels and bits add, time = slowest rank)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    import l3c_pytorch_amd  # noqa: F401
    from l3c_pytorch_amd.helpers import sharding
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = sharding.shard_indices(n_items, rank, world)
    # stand-in for the per-image work: item i has (i + 1) * 1000 pixels and costs 2 bits per sub-pixel
    pixels = sum((i + 1) * 1000 for i in mine)
    stats = {'pixels': pixels, 'subpixels': 3 * pixels, 'bits': 2.0 * 3 * pixels, 'seconds': 1.0 + rank, 'items': mine}
    everyone = sharding.gather_stats(stats)
    t = torch.tensor([stats['seconds']], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)       # what bench.py does for the timed region
    dist.barrier()
    if rank == 0:
        q.put((everyone, float(t.item())))
    dist.destroy_process_group()


def test_two_rank_sharding_and_stat_gather():
    sys.path.insert(0, ROOT)
    import l3c_pytorch_amd  # noqa: F401
    from l3c_pytorch_amd.helpers import sharding
    n_items, world = 7, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    everyone, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    items = sorted(i for s in everyone for i in s['items'])
    assert items == list(range(n_items))                              # disjoint cover
    assert everyone[0]['items'] == [0, 2, 4, 6] and everyone[1]['items'] == [1, 3, 5]
    total = sharding.combine_stats(everyone)
    assert total['pixels'] == sum((i + 1) * 1000 for i in range(n_items)) and total['ranks'] == 2
    assert total['seconds'] == 2.0 == tmax and abs(total['bpsp'] - 2.0) < 1e-12
    assert abs(total['mpix_per_s'] - total['pixels'] / 1e6 / 2.0) < 1e-12


def test_single_process_fallbacks():
    import l3c_pytorch_amd  # noqa: F401
    from l3c_pytorch_amd.helpers import sharding
    assert sharding.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    s = {'pixels': 10, 'subpixels': 30, 'bits': 60.0, 'seconds': 0.5}
    assert sharding.gather_stats(s) == [s]
    assert sharding.combine_stats([s])['bpsp'] == 2.0
