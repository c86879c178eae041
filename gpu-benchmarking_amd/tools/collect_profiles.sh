#!/bin/bash
# Standard per-round evidence for bench.py's roofline object (run ON the GPU box):
#   1. rocprofv3 --kernel-trace --stats over `python3 bench.py`           -> kernel average duration
#   2. rocprofv3 --pmc FETCH_SIZE  (own pass)                             -> HBM read bytes / launch
#   3. rocprofv3 --pmc WRITE_SIZE  (own pass: TCC has 4 slots, 3 + 2 > 4) -> HBM write bytes / launch
#   4. an un-profiled `python bench.py`
# Usage: tools/collect_profiles.sh OUTDIR     then  tools/summarize_profiles.py OUTDIR ROUND
# The program after `--` is python3 itself (no env / bash -c hop under rocprofv3).
set -u
repo="$(cd "$(dirname "$0")/../.." && pwd)"
out="${1:-$repo/gpurun_out/prof}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$repo"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra > "$out/kt_bench.json" 2> "$out/kt.err" || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.err" || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > "$out/pmc_write.json" 2> "$out/pmc_write.err" || exit 1
python3 bench.py > "$out/bench_unprofiled.json" 2> "$out/bench_unprofiled.err" || exit 1
find "$out" -name "*.db" -delete
echo profiles-done
