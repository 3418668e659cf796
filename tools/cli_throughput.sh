#!/bin/bash
# Wall-clock throughput of the command-line tools on a UHD sequence in /dev/shm (GPU box): file -> pinned buffers -> GPU -> file.
#   tools/cli_throughput.sh [frames]
n=${1:-32}
cd "$GRAFT_REPO_ROOT"
python3 - "$n" <<'PY'
import sys
sys.path.insert(0, "tests")
from synth import synth
n = int(sys.argv[1])
open("/dev/shm/vc2_in.raw", "wb").write(synth(3840, 2160, "422", 10, 1234, frames=n))
PY
E=vc2-reference_amd/bin/EncodeStream; D=vc2-reference_amd/bin/DecodeStream
args="-m HQ_ConstQ -k DD97 -d 4 -u 1 -a 2 -f 4:2:2 -x 3840 -y 2160 -l 10 -q 16 -S 2"
for dev in 0 0,0; do
  s=$(date +%s.%N); $E --devices $dev $args /dev/shm/vc2_in.raw /dev/shm/vc2_out.vc2 > /dev/null 2>&1; e=$(date +%s.%N)
  echo "EncodeStream --devices $dev: $n UHD frames in $(python3 -c "print(round($e-$s,3))") s = $(python3 -c "print(round($n/($e-$s),1))") frames/s"
  s=$(date +%s.%N); $D --devices $dev /dev/shm/vc2_out.vc2 /dev/shm/vc2_dec.raw > /dev/null 2>&1; e=$(date +%s.%N)
  echo "DecodeStream --devices $dev: $(python3 -c "print(round($n/($e-$s),1))") frames/s"
done
ls -la /dev/shm/vc2_* | awk '{print $5, $9}'
rm -f /dev/shm/vc2_*
