#!/bin/bash
# Collect the judged evidence for one round on the GPU box:  tools/profile_round.sh <tag>
# -> gpurun_out/prof_<tag>/{bench.log, trace/, pmc_fetch/, pmc_write/, pmc_sq/}; summarise with tools/summarize_profile.py
# (counters are collected in their own passes, never together with --kernel-trace: gpurun refuses such combinations)
set -e -o pipefail  # a step that fails or is killed at its limit ends the script: no further GPU step after it
tag=${1:-r}
out=gpurun_out/prof_$tag
export TMPDIR=/tmp
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 500 python3 bench.py > "$out/bench.log" 2>&1; tail -1 "$out/bench.log" | cut -c1-200
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 2 --warmup 1 --no-cpu > "$out/bench_trace.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 1 --warmup 0 --no-cpu > "$out/bench_pmc_fetch.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 1 --warmup 0 --no-cpu > "$out/bench_pmc_write.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_sq" -- python3 bench.py --steps 1 --warmup 0 --no-cpu > "$out/bench_pmc_sq.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$out/pmc_l2" -- python3 bench.py --steps 1 --warmup 0 --no-cpu > "$out/bench_pmc_l2.log" 2>&1 || echo "L2 counter pass failed (see bench_pmc_l2.log)"
python3 tools/summarize_profile.py "$out" "$out/summary" && cp "$out"/trace/*/*_kernel_stats.csv "$out/kernel_stats.csv" 2>/dev/null || true
echo "profile_round $tag done"
