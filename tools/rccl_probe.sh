C="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-rooflines --no-fresh-num-graphs"
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py $C --force-collective 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo plain; python bench.py $C 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
echo forced default queues; run
echo forced GPU_MAX_HW_QUEUES=8; GPU_MAX_HW_QUEUES=8 run
echo forced GPU_MAX_HW_QUEUES=16; GPU_MAX_HW_QUEUES=16 run
echo forced TORCH_NCCL_HIGH_PRIORITY; TORCH_NCCL_HIGH_PRIORITY=1 GPU_MAX_HW_QUEUES=8 run
echo plain GPU_MAX_HW_QUEUES=8; GPU_MAX_HW_QUEUES=8 python bench.py $C 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
