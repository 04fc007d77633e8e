#!/bin/bash
# Round-6 measurement set on one GPU box (round 5's set minus the workloads that did not change; + the driver's own command three times) (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats and the separate PMC
# passes (FETCH_SIZE | WRITE_SIZE | matrix-pipe busy | VALU / issue counters) of the workloads DESIGN.md quotes.  Output: gpurun_out/r06/.
# usage: bash tools/run_profiles.sh        (delete gpurun_out/r06/bench_line* stats_* pmc_* locally first: gpurun MERGES)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# workload tags: name:bench flags
CFGS=("c2_f32x3:" "c2_f32:--dtype f32" "c2_bf16x2:--dtype bf16x2" "c5share_bf16x2:--workload c5share --dtype bf16x2" "c5share_f32x3:--workload c5share" \
      "c3_f32x3:--workload c3" "c3_bf16x2:--workload c3 --dtype bf16x2" \
      "w512_f32x3:--width 512 --steps 60" "shipped_f32x3:--workload shipped --steps 30 --warmup 5" \
      "c4strong1_f32x3:--scaling strong --steps 60" "c2_f32x3_classic:")
# the driver's command (bench.py --gpus 1 --steps 20 --warmup 5), three times on this box
for i in 1 2 3; do $B --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_cmd_$i.json 2>/dev/null; done
# ---- bench lines (default flags; the headline one with the CPU baseline)
for cfg in "${CFGS[@]}"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  if [ "$tag" = "c2_f32x3" ]; then $B > $O/bench_line_$tag.json 2> $O/bench_line_$tag.err;
  elif [ "$tag" = "c2_f32x3_classic" ]; then MVSDF_DEFERRED_STEP=0 $B --no-cpu-baseline > $O/bench_line_$tag.json 2>/dev/null;    # the classic step (one host wait per forward): the A/B partner of the deferred default
  else $B --no-cpu-baseline $args > $O/bench_line_$tag.json 2>/dev/null; fi
done
# the collective path at world size 1 (RCCL): what the process-group hand-off costs per step
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 1 --no-cpu-baseline > $O/bench_line_c2_f32x3_rccl1.json 2>/dev/null
# ---- kernel stats
PROF=("c2_f32x3:" "c5share_bf16x2:--workload c5share --dtype bf16x2" "c3_f32x3:--workload c3" "shipped_f32x3:--workload shipped")
for cfg in "${PROF[@]}"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- $B --steps 20 --warmup 3 --no-cpu-baseline $args > /dev/null 2>&1
done
python3 $R/tools/trace_timeline.py $O/stats_c2_f32x3 15 > $O/step_timeline_c2_f32x3.txt 2>&1
python3 $R/tools/trace_timeline.py $O/stats_c3_f32x3 15 > $O/step_timeline_c3_f32x3.txt 2>&1
# ---- PMC passes (each its own run; no trace domains besides --kernel-trace)
for cfg in "${PROF[@]}"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_$tag/fetch --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_$tag/write --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $O/pmc_$tag/mfma --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS -d $O/pmc_$tag/valu --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
done
# keep the merge small: the per-dispatch CSVs only
find $O -name "*agent_info.csv" -delete
du -sh $O
head -c 300 $O/bench_line_c2_f32x3.json
