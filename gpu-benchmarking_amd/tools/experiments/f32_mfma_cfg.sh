#!/bin/bash
# fp32 2D nq 17..32: matrix-core configurations (SF_F32_MFMA_CFG, a development knob of bwdtrans_quad.hip) against the
# vector kernel (cfg 5).  Usage: f32_mfma_cfg.sh OUTFILE
out=${1:-gpurun_out/f32_mfma_cfg.log}
: > $out
for cfg in 0 1 2 3 4 5; do
  echo "== SF_F32_MFMA_CFG=$cfg (0: EC2 MINW2 K2, 1: EC2 MINW4 K2, 2: EC4 MINW2 K1, 3: EC4 MINW4 K1, 4: EC2 MINW4 K1, 5: vector kernel)" >> $out
  SF_F32_MFMA_CFG=$cfg python3 gpu-benchmarking_amd/tools/sweep_auto.py 1048576 10 f32 2>/dev/null | grep -E "^2D nq(1[7-9]|2[0-9]|3[0-2]) " >> $out
done
