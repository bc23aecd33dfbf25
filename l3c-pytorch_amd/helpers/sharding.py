"""Multi-GPU = replicas only: images (or auto-crops) are independent, every rank codes its own share
with its own copy of the 20 MB model, and there is no collective on the data path.  The only communication is a host-side
gather of per-rank statistics, done with whatever process group is active (RCCL on GPUs, gloo in the CPU tests)."""
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """Static round-robin: item i goes to rank i % world_size."""
    return list(range(rank, n_items, world_size))


def shard_balanced(costs, rank, world_size):
    """Largest-first greedy assignment (LPT): items in order of decreasing cost, each to the rank with the least cost so far (ties:
    the lower index, the lower rank) -- every rank computes the same deterministic plan from the same `costs`, so there is still no
    communication.  For sets whose items differ in size (config 4: image areas differ 4x) the slowest rank sets the aggregate rate;
    round robin leaves it ~10 % above the mean at 8 ranks x 62 images, this within 1-2 % (tests/test_host_logic.py).
    -> the rank's item indices, in increasing order."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world_size
    mine = []
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        load[r] += costs[i]
        if r == rank:
            mine.append(i)
    return sorted(mine)


def host_budget(world_size, cpus=None):
    """Per-rank share of ONE host that runs world_size ranks (one process per GPU): intra-op CPU threads, I/O worker threads and
    page-locked staging buffers, so that 8 ranks do not each claim what one rank may (256 hardware threads, eight 32 MB pinned
    buffers, 8 decode threads).  Pure function of (world, cpus)."""
    import os
    cpus = cpus or os.cpu_count() or 1
    per = max(1, cpus // max(world_size, 1))
    return {'torch_threads': max(1, min(16, per)), 'io_threads': max(1, min(8, per // 4 or 1)),
            'pinned_buffers': 8 if world_size <= 2 else 4 if world_size <= 4 else 3}


def gather_stats(stats):
    """stats: dict of python numbers -> list of every rank's dict (on every rank); [stats] without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return [stats]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, stats)
    return out


def combine_stats(all_stats):
    """Aggregate throughput / bpsp over ranks: pixels and bits add up, time is the slowest rank's (weak scaling)."""
    pixels = sum(s['pixels'] for s in all_stats)
    bits = sum(s['bits'] for s in all_stats)
    subpixels = sum(s['subpixels'] for s in all_stats)
    seconds = max(s['seconds'] for s in all_stats)
    return {'pixels': pixels, 'seconds': seconds, 'mpix_per_s': pixels / 1e6 / seconds if seconds > 0 else float('inf'),
            'bpsp': bits / subpixels if subpixels else float('nan'), 'ranks': len(all_stats)}
