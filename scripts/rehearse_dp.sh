#!/bin/bash
# Rehearses the multi-rank code paths of bench.py on one GPU (control flow, not a measurement).
set -e
F="--no-cpu-baseline --no-roofline --steps 6 --warmup 3"
echo "== one rank, RCCL communicator, all-reduce between the tail graphs"
timeout -k 10 200 python bench.py $F --rehearse-backend nccl | cut -c1-400
echo "== one rank, RCCL communicator, reduce-scatter mode"
timeout -k 10 200 python bench.py $F --rehearse-backend nccl --dp-mode rs_ag | cut -c1-400
echo "== two ranks on one GPU over gloo"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 $F --rehearse-backend gloo | cut -c1-400
echo "== two ranks on one GPU over gloo, reduce-scatter mode"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 $F --rehearse-backend gloo --dp-mode rs_ag | cut -c1-400
