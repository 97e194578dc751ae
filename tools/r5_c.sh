#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_gpu_facade.py tests/test_gpu_bench_two_ranks.py -m gpu -q -x -k "allgather or sharded or two_ranks" > $O/pytest_c.txt 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|Gloo" $O/pytest_c.txt | tail -40
