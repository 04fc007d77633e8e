#!/bin/bash
# Round-3 measurement set on one GPU box (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats and the separate PMC
# passes (FETCH_SIZE | WRITE_SIZE | matrix-pipe busy | VALU / issue counters) of the workloads DESIGN.md quotes.  Output: gpurun_out/r03/.
# usage: bash tools/run_profiles.sh
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# ---- bench lines (default flags; the headline one with the CPU baseline)
$B > $O/bench_line.json 2> $O/bench_line.err
$B --no-cpu-baseline --dtype bf16 > $O/bench_line_c2_bf16.json 2>/dev/null
$B --no-cpu-baseline --dtype bf16 --workload c5share > $O/bench_line_c5share_bf16.json 2>/dev/null
$B --no-cpu-baseline --workload c5share > $O/bench_line_c5share_f32.json 2>/dev/null
$B --no-cpu-baseline --workload c3 > $O/bench_line_c3_f32.json 2>/dev/null
$B --no-cpu-baseline --workload c3 --dtype bf16 > $O/bench_line_c3_bf16.json 2>/dev/null
$B --no-cpu-baseline --width 512 --steps 60 > $O/bench_line_w512_f32.json 2>/dev/null
# ---- kernel stats
for cfg in "c2_f32:" "c5share_bf16:--workload c5share --dtype bf16" "c3_f32:--workload c3" "w512_f32:--width 512"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- $B --steps 20 --warmup 3 --no-cpu-baseline $args > /dev/null 2>&1
done
# ---- PMC passes (each its own run; no trace domains besides --kernel-trace)
for cfg in "c2_f32:" "c5share_bf16:--workload c5share --dtype bf16" "c3_f32:--workload c3"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_$tag/fetch --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_$tag/write --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $O/pmc_$tag/mfma --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS -d $O/pmc_$tag/valu --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
done
# keep the merge small: the per-dispatch CSVs only
find $O -name "*agent_info.csv" -delete
du -sh $O
head -c 300 $O/bench_line.json
