#!/bin/bash
# Sweep driver, counterpart of the reference's benchmark0N/run.sh (benchmark05/run.sh:1-8):
# one log per order, named nq{N}x{N}x{N}.log (3D) / nq{N}x{N}.log (2D), stdout+stderr captured,
# plus outfile.log for benchmark01/02/03 (each in its own directory, as the reference keeps them).
#   tools/run.sh [outdir] [hex|quad|misc|all]
# HIP_VISIBLE_DEVICES selects the GPU (the reference used CUDA_VISIBLE_DEVICES=1).
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-$here/results}"
what="${2:-all}"
mkdir -p "$out/benchmark01" "$out/benchmark02" "$out/benchmark03" "$out/benchmark04" "$out/benchmark05"
if [ "$what" = hex ] || [ "$what" = all ]; then
  for i in 2 3 4 5 6 7 8 9 10; do
    echo "hex nq=$i"; "$here/bin/benchmark05" $i $i $i --json "$out/benchmark05/nq${i}x${i}x${i}.json" &> "$out/benchmark05/nq${i}x${i}x${i}.log"
  done
fi
if [ "$what" = quad ] || [ "$what" = all ]; then
  for i in 2 4 6 8 10 12 14 16 20 24 26 28 30 32; do
    echo "quad nq=$i"; "$here/bin/benchmark04" $i $i --json "$out/benchmark04/nq${i}x${i}.json" &> "$out/benchmark04/nq${i}x${i}.log"
  done
fi
if [ "$what" = misc ] || [ "$what" = all ]; then
  "$here/bin/benchmark01" &> "$out/benchmark01/outfile.log"
  "$here/bin/benchmark02" &> "$out/benchmark02/outfile.log"
  "$here/bin/benchmark03" &> "$out/benchmark03/outfile.log"
fi
echo sweep-done
