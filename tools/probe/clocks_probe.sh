#!/bin/bash
# what this box offers for reading the GPU's clocks / power without root (bench.py's `clocks` block is built on it)
for d in /sys/class/drm/card*/device; do
  echo "== $d"; ls $d | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk power_dpm_force_performance_level current_compute_partition current_memory_partition gpu_busy_percent mem_busy_percent; do
    [ -r $d/$f ] && { echo "-- $f"; cat $d/$f; }
  done
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input freq1_input freq2_input temp1_input; do [ -r $h/$f ] && echo "$f: $(cat $h/$f)"; done; done
done
rocm-smi --showclocks --showpower --json 2>&1 | head -c 3000; echo
(time rocm-smi --showclocks --showpower --json >/dev/null) 2>&1 | tail -3
amd-smi metric -c -p --json 2>&1 | head -c 3000; echo
