"""Multi-GPU = replicas only: images (or auto-crops) are independent, every rank codes its own share
with its own copy of the 20 MB model, and there is no collective on the data path.  The only communication is a host-side
gather of per-rank statistics, done with whatever process group is active (RCCL on GPUs, gloo in the CPU tests)."""
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """Static round-robin: item i goes to rank i % world_size."""
    return list(range(rank, n_items, world_size))


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
