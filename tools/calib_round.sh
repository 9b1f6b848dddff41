#!/bin/bash
# FETCH_SIZE calibration on the GPU box (tools/micro/fetch_calib.hip): one timing run + one counter pass per table size.
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/calib
rm -rf "$out"; mkdir -p "$out"
for lg in 24 27 31; do
  timeout -k 10 120 ./tools/micro/fetch_calib $lg > "$out/time_$lg.log" 2>&1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_$lg" -- ./tools/micro/fetch_calib $lg > "$out/pmc_$lg.log" 2>&1
done
grep -h CALIB "$out"/time_*.log | tail -24
echo calib done
