#!/bin/bash
# Wall-clock throughput of the command-line tools on a UHD sequence in /dev/shm (GPU box): file -> pinned buffers -> GPU -> file.
#   tools/cli_throughput.sh [frames] [device lists ...]      default: 256 frames; "0" "0,0" "0,0,0,0"
# Per run: the whole process (start-up, context creation, first touch of the output pages included) and the steady state the
# tools report themselves with VC2_TOOL_STATS=1 (pictures per second from the completion of picture 4 x workers to the last).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
n=${1:-256}; shift
devs=${@:-0 0,0 0,0,0,0}
python3 - "$n" <<'PY'
import sys
sys.path.insert(0, "tests")
from synth import synth
n = int(sys.argv[1])
few = synth(3840, 2160, "422", 10, 1234, frames=8)     # eight distinct frames, repeated (numpy makes 0.45 frames per second)
with open("/dev/shm/vc2_in.raw", "wb") as f:
    for k in range((n + 7) // 8): f.write(few)
PY
E=vc2-reference_amd/bin/EncodeStream; D=vc2-reference_amd/bin/DecodeStream
args="-m HQ_ConstQ -k DD97 -d 4 -u 1 -a 2 -f 4:2:2 -x 3840 -y 2160 -l 10 -q 16 -S 2"
export VC2_TOOL_STATS=1
for dev in $devs; do
  s=$(date +%s.%N); $E --devices $dev $args /dev/shm/vc2_in.raw /dev/shm/vc2_out.vc2 2> /dev/shm/vc2_e.err > /dev/null; e=$(date +%s.%N)
  echo "--devices $dev EncodeStream: whole process $(python3 -c "print(round($n/($e-$s),1))") frames/s;  $(grep stats /dev/shm/vc2_e.err)"
  rm -f /dev/shm/vc2_dec.raw
  s=$(date +%s.%N); $D --devices $dev /dev/shm/vc2_out.vc2 /dev/shm/vc2_dec.raw 2> /dev/shm/vc2_d.err > /dev/null; e=$(date +%s.%N)
  echo "--devices $dev DecodeStream (new output file): whole process $(python3 -c "print(round($n/($e-$s),1))") frames/s;  $(grep stats /dev/shm/vc2_d.err)"
  # again over the file of the first run: its pages exist (a NEW file in /dev/shm is filled at 4 - 6 GB/s however many threads
  # write it -- tools/probe/pagetouch.c -- which is 120 - 190 UHD frames/s whatever the decoder does).  Round 5: the tool empties
  # an existing output file up front unless VC2_DECODESTREAM_REUSE=1 asks for the reuse (a killed run must not leave old frames)
  s=$(date +%s.%N); VC2_DECODESTREAM_REUSE=1 $D --devices $dev /dev/shm/vc2_out.vc2 /dev/shm/vc2_dec.raw 2> /dev/shm/vc2_d.err > /dev/null; e=$(date +%s.%N)
  echo "--devices $dev DecodeStream (over an existing file): whole process $(python3 -c "print(round($n/($e-$s),1))") frames/s;  $(grep stats /dev/shm/vc2_d.err)"
  [ "$dev" = "0" ] && cp /dev/shm/vc2_out.vc2 /dev/shm/vc2_out0.vc2 && sha256sum /dev/shm/vc2_dec.raw | cut -c1-16 > /dev/shm/vc2_dec0.sha
  cmp -s /dev/shm/vc2_out.vc2 /dev/shm/vc2_out0.vc2 || echo "STREAMS DIFFER between --devices 0 and $dev"
  [ "$(sha256sum /dev/shm/vc2_dec.raw | cut -c1-16)" = "$(cat /dev/shm/vc2_dec0.sha)" ] || echo "DECODED FILES DIFFER between --devices 0 and $dev"
done
ls -la /dev/shm/vc2_* | awk '{print $5, $9}'
rm -f /dev/shm/vc2_*
