#!/bin/bash
# Counter evidence for the bandwidth -> compute crossover across the order sweep (run ON the GPU box):
#   for each (benchmark, nq, variant): kernel trace (duration) + three PMC passes
#   (SQ/GRBM compute counters, FETCH_SIZE, WRITE_SIZE: separate passes, TCC has 4 slots).
# Usage:  tools/profile_crossover.sh OUTDIR [FILTER]   (then tools/summarize_crossover.py OUTDIR)
#         FILTER = grep -E pattern on the configuration tags (default: all)
# The program after `--` is the driver binary itself (no env/bash -c hop under rocprofv3).
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-gpurun_out/crossover}"
filter="${2:-.}"
mkdir -p "$out"
out="$(cd "$out" && pwd)"
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
run() { # tag exe args...
  local tag="$1"; shift
  echo "$tag" | grep -Eq "$filter" || return 0
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag/kt" -- "$@" > "$out/$tag.log" 2>&1 || return 1
  rocprofv3 --pmc $SQ --output-format csv -d "$out/$tag/sq" -- "$@" > /dev/null 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/$tag/fetch" -- "$@" > /dev/null 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/$tag/write" -- "$@" > /dev/null 2>&1 || return 1
  echo "$tag ok"
}
N=1048576
M=262144   # element count for the 3D orders above 10 (their elements are 20-60 KB)
for nq in 2 3 4 5 6 7 8 9 10; do
  run hex_nq${nq}_wave "$here/bin/benchmark05" $nq $nq $nq --nelmt $N --no-baselines --data random --variant wave || exit 1
done
run hex_nq11_wave "$here/bin/benchmark05" 11 11 11 --nelmt $M --no-baselines --data random --variant wave || exit 1
for nq in 8 16 20 24 32; do
  run quad_nq${nq}_wave "$here/bin/benchmark04" $nq $nq --nelmt $N --no-baselines --data random --variant wave || exit 1
done
for nq in 8 10; do
  run hex_nq${nq}_mfma "$here/bin/benchmark05" $nq $nq $nq --nelmt $N --no-baselines --data random --variant mfma || exit 1
done
for nq in 12 14 16; do
  run hex_nq${nq}_mfma "$here/bin/benchmark05" $nq $nq $nq --nelmt $M --no-baselines --data random --variant mfma || exit 1
done
for nq in 12 16 20 24 28 32; do
  run quad_nq${nq}_mfma "$here/bin/benchmark04" $nq $nq --nelmt $N --no-baselines --data random --variant mfma || exit 1
done
# round 2: the 4x4x4_4b matrix-core kernel (AUTO at nq 21..31)
for nq in 16 20 22 24 26 28 30 32; do
  run quad_nq${nq}_mfma4 "$here/bin/benchmark04" $nq $nq --nelmt $N --no-baselines --data random --variant mfma4 || exit 1
done
# round 3: anisotropic extents through SF_VARIANT_AUTO (compile-time triples, bwdtrans_wave3.h) and the run-time-extent kernel
for shape in "8 8 4" "4 8 6" "10 6 8"; do
  set -- $shape
  run hex_$1x$2x$3_auto "$here/bin/benchmark05" $1 $2 $3 --nelmt $N --no-baselines --data random --variant auto || exit 1
  run hex_$1x$2x$3_wave-rt "$here/bin/benchmark05" $1 $2 $3 --nelmt $N --no-baselines --data random --variant wave-rt || exit 1
done
# round 3: the 3D 4x4x4_4b matrix-core kernel (bwdtrans_hmfma4.h; AUTO at nq 12 and 16)
for nq in 12 14 16; do
  run hex_nq${nq}_mfma4 "$here/bin/benchmark05" $nq $nq $nq --nelmt $M --no-baselines --data random --variant mfma4 || exit 1
done
echo all-done
