#!/usr/bin/env python
"""bench.py -- MPix/s of the L3C encode hot path (net forward + fused logistic-mixture CDF head + HIP range coder) on
synthetic 768x512 RGB batches, BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of B images per GPU: the images are already resident in HBM when the
timed region starts, and the step ends with every stream's bytes and byte counts in HBM (file assembly / PCIe are not
part of `value`).  Images shard one-batch-per-GPU with no data-path collective ("replicas only",
SURVEY.md section 8e): scaling is weak, `value` = all ranks' pixels / max-over-ranks time.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline      dominant kernel = the fp32 MFMA conv (v_mfma_f32_32x32x2_f32; Winograd F(2x2,3x3) for the 3x3 layers):
                algorithmic (direct-convolution) FLOPs of all its launches in the timed region / their summed HIP-event
                durations, against the 157.3 TFLOP/s dense fp32 MFMA peak (+ the executed share of the matrix pipe)
  cpu_baseline  the oracle (CPU restatement of the reference path: torch-CPU convs + torch CDF tables + C range coder)
                timed on this box's host cores on ONE 768x512 image of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

H, W = 512, 768
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
ALGO_FLOP_PER_PX = 1367796         # conv stack of the L3C forward, FLOP per image pixel


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=128, help='images per GPU per step')
    ap.add_argument('--coder-cus', type=int, default=0, help='compute units reserved for the range coder (0 = share all CUs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-decode', action='store_true', help='skip the (untimed-region) decode leg')
    ap.add_argument('--no-kernel-events', action='store_true', help='skip the per-launch HIP events of the roofline leg')
    return ap.parse_args()


def build_path(device, coder_cus=0):
    import l3c_pytorch_amd  # noqa: F401
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint
    from l3c_pytorch_amd.helpers import config_parser, synthetic
    cfg = config_parser.parse_builtin('ms', 'cr')
    sd = synthetic.make_state_dict(cfg, 0)
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(sd, strict=True)
    bp.set_eval()
    return cfg, sd, bp, Bitcoding(bp, coder_cus=coder_cus), synthetic


def cpu_baseline(sd, synthetic):
    """The oracle's encode of one 768x512 image on the host cores (bounded: one image, ~10-30 s)."""
    from oracle import bitcoding as obc
    img = synthetic.make_image(H, W, 0, 'natural').unsqueeze(0).long()
    cores = torch.get_num_threads()
    t0 = time.time()
    with torch.no_grad():
        data = obc.encode(img, sd)
    dt = time.time() - t0
    return {'value': round(H * W / 1e6 / dt, 5), 'unit': 'MPix/s', 'cores': cores, 'kind': 'port',
            'sample': '1 image 768x512 (natural-like synthetic), oracle.bitcoding.encode: torch-CPU forward + torch CDF '
                      'tables + C range coder, {:.1f} s, {} bytes'.format(dt, len(data)),
            'seconds': round(dt, 2)}


def decode_leg(bc, enc, imgs, compute_stream):
    """Secondary figure (SURVEY.md section 8d "also decode MPix/s"), outside the timed region: decode the last coded batch
    back from its `.l3c` byte strings (host) to pixels in HBM and check it is lossless.  The range decoder is a serial
    chain per stream (one wavefront each; the R, G, B chains of an image run a chunk of pixels apart), so the time depends
    little on the batch size."""
    SYMBOLS_PER_PX = 4.640625           # 3 P0 + 5 (P1 + P2 + P3) symbols per image pixel, SURVEY.md section 8d
    with torch.cuda.stream(compute_stream):
        files = enc.to_bytes()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec, _ = bc.decode_batch(files)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lossless = bool(torch.equal(dec.to(torch.uint8), imgs.to(torch.uint8)))
    B = len(files)
    return {'value': round(B * H * W / 1e6 / dt, 3), 'unit': 'MPix/s', 'batch': B, 'seconds': round(dt, 3),
            'lossless': lossless, 'msym_per_s_aggregate': round(B * H * W * SYMBOLS_PER_PX / 1e6 / dt, 1),
            'longest_chain_symbols': H * W,
            'note': 'host .l3c bytes -> pixels in HBM; latency-bound: serial chains of {} symbols per RGB channel, the three '
                    'channels pipelined a chunk of pixels apart'.format(H * W)}


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    # launched by torch.distributed.run (any world size, 1 included): RCCL for the barrier and the max-over-ranks time
    distributed = world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)   # RCCL; one process per GPU
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus {} != WORLD_SIZE {}'.format(args.gpus, world), file=sys.stderr)

    cfg, sd, bp, bc, synthetic = build_path(device, args.coder_cus)
    from l3c_pytorch_amd import _lib, ops
    B = args.batch
    # synthetic images (seed = global image index), resident in HBM before the timed region
    imgs = torch.stack([synthetic.make_image(H, W, rank * B + i, 'natural') for i in range(B)]).to(device)
    imgs_f = imgs.float().contiguous()
    torch.cuda.synchronize()

    compute_stream = bc.compute_stream if bc.compute_stream is not None else torch.cuda.current_stream()

    def step():
        with torch.cuda.stream(compute_stream):
            out = bp.net(imgs_f)
            return bc.encode_batch(imgs_f, out=out)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        enc = step()
    barrier()
    want_events = rank == 0 and not args.no_kernel_events
    ops.PROFILE = [] if want_events else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        enc = step()
    barrier()
    elapsed = time.perf_counter() - t0
    records, ops.PROFILE = ops.PROFILE, None

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_px = world * B * args.steps * H * W
    value = total_px / 1e6 / elapsed

    # bpsp of the coded batch (file bytes incl. framing) and lossless check of one image on rank 0
    with torch.cuda.stream(compute_stream):
        sizes = enc.file_sizes().cpu().numpy()
    bpsp = float(sizes.sum()) * 8 / (B * 3 * H * W)

    result = None
    if rank == 0:
        roofline = None
        if records:
            by = {}
            for key, flops, e0, e1 in records:
                d = by.setdefault(key, [0.0, 0.0, 0])
                d[0] += flops
                d[1] += e0.elapsed_time(e1) * 1e-3
                d[2] += 1
            dom = max(by, key=lambda k: by[k][1])
            flops, secs, n = by[dom]
            all_f = sum(v[0] for v in by.values())
            all_s = sum(v[1] for v in by.values())
            traffic = None   # HBM bytes per launch of the dominant kernel, from the committed PMC passes (profiles/)
            try:
                with open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic_conv3x3.json')) as f:
                    traffic = int(json.load(f)['hbm_bytes_per_flop'] * flops / n)
            except (OSError, KeyError, ValueError):
                pass
            # the Winograd kernel executes 16 multiplications per 2x2 output tile and channel pair where the direct form needs
            # 36: `achieved` stays ALGORITHMIC (direct-convolution) FLOPs / time, so it can exceed the MFMA peak; the share
            # of the matrix pipe really in use is reported next to it
            executed = 16.0 / 36.0 if dom.startswith('conv_wino') else 1.0
            roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': round(flops / secs / 1e12, 2),
                        'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(flops / secs / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                        'mfma_executed_tflops': round(flops * executed / secs / 1e12, 2),
                        'mfma_utilisation': round(flops * executed / secs / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                        'note': ('Winograd F(2x2,3x3): achieved = algorithmic (direct-conv) FLOPs / time; the MFMA executes 16/36 of '
                                 'them' if executed < 1 else 'implicit GEMM: algorithmic = executed FLOPs'),
                        'traffic': traffic, 'launches': n, 'avg_launch_us': round(secs / n * 1e6, 2),
                        'algorithmic_gflop_per_launch': round(flops / n / 1e9, 3),
                        'all_mfma_convs': {'achieved': round(all_f / all_s / 1e12, 2), 'seconds_per_step': round(all_s / args.steps, 5),
                                           'share_of_step': round(all_s / elapsed, 3)},
                        'per_kernel': {k: {'tflops': round(v[0] / v[1] / 1e12, 1), 'share_of_step': round(v[1] / elapsed, 3),
                                           'launches_per_step': v[2] // args.steps} for k, v in sorted(by.items())}}
        decode = None
        if world == 1 and not args.no_decode:
            decode = decode_leg(bc, enc, imgs, compute_stream)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sd, synthetic)
        name, ncu, arch = _lib.device_info()
        result = {
            'metric': 'MPix/s encode (net+CDF+AC) on 768x512 RGB', 'value': round(value, 3), 'unit': 'MPix/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'L3C 0306_0001 (cr.cf, synthetic seeded checkpoint), batch of 768x512 synthetic RGB per '
                                   'GPU: net forward + fused logistic-mixture CDF head + HIP range coder -> bytes in HBM',
                       'batch_per_gpu': B, 'image': '768x512', 'coder_cus': args.coder_cus, 'sharding': 'images, replicas only (no collective)'},
            'bpsp': round(bpsp, 4), 'flop_per_px': ALGO_FLOP_PER_PX,
            'end_to_end_tflops': round(value * 1e6 * ALGO_FLOP_PER_PX / 1e12 / world, 2),
            'device': '{} ({}, {} CUs)'.format(name, arch, ncu),
            'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 1e9, 1),
            'roofline': roofline, 'cpu_baseline': cpu, 'decode': decode,
        }
        print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == '__main__':
    main()
