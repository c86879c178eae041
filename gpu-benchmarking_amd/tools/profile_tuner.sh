#!/bin/bash
# Per-kernel counters of every variant a tuner binary runs (run ON the GPU box): kernel trace + two SQ passes.
# Usage: tools/profile_tuner.sh OUTDIR TUNER [args...]   then tools/summarize_tuner_profile.py OUTDIR
# The program after `--` is the tuner binary itself (no env / bash -c hop under rocprofv3).
set -u
out="$1"; shift
mkdir -p "$out"
out="$(cd "$out" && pwd)"
exe="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; shift
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- "$exe" "$@" > "$out/tuner.log" 2>&1 || exit 1
rocprofv3 --pmc $P1 --output-format csv -d "$out/p1" -- "$exe" "$@" > /dev/null 2>&1 || exit 1
rocprofv3 --pmc $P2 --output-format csv -d "$out/p2" -- "$exe" "$@" > /dev/null 2>&1 || exit 1
echo "profile_tuner ok: $out"
