#!/bin/bash
# Issue / stall counters of the flagship hex kernel (run ON the GPU box): two SQ passes per precision.
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-gpurun_out/stalls}"
mkdir -p "$out"
out="$(cd "$out" && pwd)"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_UNALIGNED_STALL"
for p in f64 f32; do
  for n in 1 2 3; do
    eval "C=\$P$n"
    rocprofv3 --pmc $C --output-format csv -d "$out/${p}_p$n" -- "$here/bin/benchmark05" 8 8 8 --nelmt 1048576 --no-baselines --data random --precision $p > /dev/null 2>&1 || echo "pass $p $n failed"
  done
done
echo done
