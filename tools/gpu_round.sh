#!/bin/bash
# One GPU-box call: GPU tests, the default bench line, the 3-rank self-launch rehearsal.  tools/gpu_round.sh <tag>
set -e -o pipefail
tag=${1:-r}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$tag.log 2>&1 || { tail -30 gpurun_out/pytest_$tag.log; exit 1; }
tail -3 gpurun_out/pytest_$tag.log
timeout -k 10 400 python bench.py > gpurun_out/bench_$tag.log 2>&1 || { tail -30 gpurun_out/bench_$tag.log; exit 1; }
tail -1 gpurun_out/bench_$tag.log | cut -c1-600
