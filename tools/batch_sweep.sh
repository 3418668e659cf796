#!/bin/bash
# tools/batch_sweep.sh: pictures per step of the other BASELINE configurations (tools/time_cfg.py, release library, one box):
# what bench.py's other_configs should run each configuration at
for c in cfg1@16 cfg1@32 cfg1@64 cfg1@128 cfg1@256 cfg3@16 cfg3@32 cfg3@64 cfg3@128 cfg4@4 cfg4@8 cfg4@16 cfg4@32 cfg5@16 cfg5@32 cfg5@64 cfg5@128 cfg5@256 cfg2@128; do
  echo "$(python tools/time_cfg.py $c 2>&1 | grep -v amdgpu)"
done
