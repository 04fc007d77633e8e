cd $GRAFT_REPO_ROOT
for lib in "" pd22 pd24; do
  if [ -n "$lib" ]; then export MVSDF_LIB=$PWD/mvsdf_amd/libmvsdf_hip_$lib.so; else unset MVSDF_LIB; fi
  echo "== lib ${lib:-product (PD2=0)}"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$lib && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$lib -- python3 $GRAFT_REPO_ROOT/tools/micro/chain_x3/fwd_ab.py 256 6200 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/tools/ktrace_medians.py /tmp/prof_$lib k_chain_fwd_x3)
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profb_$lib && rocprofv3 --kernel-trace --output-format csv -d /tmp/profb_$lib -- python3 $GRAFT_REPO_ROOT/tools/micro/chain_x3/fwd_ab.py 256 12400 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/tools/ktrace_medians.py /tmp/profb_$lib k_chain_fwd_x3)
done
