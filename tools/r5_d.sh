#!/bin/bash
# round 5: the pruned tree against the tree before (frames byte for byte on C3 / C4 / C5 states), then the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
P=$PWD/gym-duckietown_amd/lib/libdtsim_prev.so
{
for cfg in "MAP=small_loop DR=0" "MAP=loop_pedestrians DR=1 STEPS=60" "MAP=loop_only_duckies,small_loop_only_duckies DR=0"; do
  env $cfg N=512 DTSIM_LIB=$P python tools/lib_frames.py prev 2>&1 | tail -1
  env $cfg N=512 python tools/lib_frames.py cur 2>&1 | tail -1
  python tools/lib_frames.py prev cur 2>&1 | tail -1
done
} > $O/prune_frames.txt 2>&1
cat $O/prune_frames.txt
python -m pytest tests -m gpu -q -x > $O/pytest_d.txt 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|Gloo" $O/pytest_d.txt | tail -6
bash tools/ab_cfg.sh "c3 c4 c5" prev default > $O/ab_prune.txt 2>&1; cat $O/ab_prune.txt
