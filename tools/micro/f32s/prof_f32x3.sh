set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/f32x3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
for cfg in "c2_f32x3:--dtype f32x3" "c3_f32x3:--workload c3 --dtype f32x3"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- $B --steps 20 --warmup 3 --no-cpu-baseline $args > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $O/pmc_$tag/mfma --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS -d $O/pmc_$tag/valu --output-format csv -- $B --steps 5 --warmup 2 --no-cpu-baseline $args > /dev/null 2>&1
done
find $O -name "*agent_info.csv" -delete
du -sh $O
