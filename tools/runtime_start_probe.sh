#!/bin/bash
# usage (GPU box): tools/runtime_start_probe.sh -> the HIP runtime's start under a few environments, three runs each
P=tools/ubench/runtime_start
ls /sys/class/kfd/kfd/topology/nodes/ 2>/dev/null | tr '\n' ' '; echo "<- kfd topology nodes"
nproc
for envs in "" "ROCR_VISIBLE_DEVICES=0" "HIP_VISIBLE_DEVICES=0" "HSA_ENABLE_SDMA=0" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=1" "HSA_NO_SCRATCH_RECLAIM=1" "HIP_HOST_COHERENT=0"; do
  echo "== env: ${envs:-default}"
  for i in 1 2 3; do env $envs $P 536870912 | tr -s ' ' | tr '\n' ';'; echo; done
done
