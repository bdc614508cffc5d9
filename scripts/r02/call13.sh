#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 5 --warmup 2 --batch 2048 --backend gloo --share-device 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -5
