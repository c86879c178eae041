#!/bin/bash
# Sweep driver, counterpart of the reference's benchmark0N/run.sh (benchmark05/run.sh:1-8):
# one log per order, named nq{N}x{N}x{N}.log (3D) / nq{N}x{N}.log (2D), stdout+stderr captured.
#   tools/run.sh [outdir]      HIP_VISIBLE_DEVICES selects the GPU (the reference used CUDA_VISIBLE_DEVICES=1)
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-$here/logs}"
mkdir -p "$out"
for i in 2 3 4 5 6 7 8 9 10; do
  echo "hex nq=$i"; "$here/bin/benchmark05" $i $i $i &> "$out/nq${i}x${i}x${i}.log"
done
for i in 2 4 6 8 10 12 14 16 32; do
  echo "quad nq=$i"; "$here/bin/benchmark04" $i $i &> "$out/nq${i}x${i}.log"
done
"$here/bin/benchmark01" &> "$out/outfile.log"
