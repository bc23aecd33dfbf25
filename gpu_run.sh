export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-decode 2>&1 | tail -2 | cut -c1-400
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/bench_dataset.py --images 16 2>&1 | tail -3
