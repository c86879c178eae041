#!/bin/bash
# HBM traffic per launch for every shape bench.py reports (run ON the GPU box): the 3D order sweep 2..10 and the 2D
# orders of bench.py's quad list at 1 Mi elements, and the flagship at 1.25 M / 10 M elements (BASELINE configs[4]:
# one GPU's shard of the 8-GPU job, and the whole batch on one GPU).  Two PMC passes per shape -- FETCH_SIZE and
# WRITE_SIZE cannot share a pass (TCC has 4 counter slots) -- each over the C++ driver with the flagship column only.
# Usage: tools/collect_traffic.sh OUTDIR   then   tools/summarize_traffic.py OUTDIR ROUND   (repo root)
# The program after `--` is the driver binary itself (no env / bash -c hop under rocprofv3).
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-gpurun_out/traffic}"
mkdir -p "$out"
out="$(cd "$out" && pwd)"
cd /tmp && export TMPDIR=/tmp
run() { # tag exe args...
  local tag="$1"; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/$tag/fetch" -- "$@" > "$out/$tag.log" 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/$tag/write" -- "$@" > /dev/null 2>&1 || return 1
  find "$out/$tag" -name "*.db" -delete
  echo "$tag ok"
}
N=1048576
for nq in 2 3 4 5 6 7 8 9 10; do
  run hex_${nq}_$N "$here/bin/benchmark05" $nq $nq $nq --nelmt $N --no-baselines --data random || exit 1
done
for nq in 2 4 6 8 10 12 14 16 20 24 26 28 30 32; do
  run quad_${nq}_$N "$here/bin/benchmark04" $nq $nq --nelmt $N --no-baselines --data random || exit 1
done
# anisotropic extents (round 3): the compile-time triples of csrc/bwdtrans_wave3.h and one run-time-extent shape
for shape in "8 8 4" "4 8 6" "10 6 8" "3 5 4"; do
  set -- $shape
  run hex_$1x$2x$3_$N "$here/bin/benchmark05" $1 $2 $3 --nelmt $N --no-baselines --data random || exit 1
done
run hex_8_1250000 "$here/bin/benchmark05" 8 8 8 --nelmt 1250000 --no-baselines --data random || exit 1
run hex_8_10000000 "$here/bin/benchmark05" 8 8 8 --nelmt 10000000 --no-baselines --data random || exit 1
echo traffic-done
