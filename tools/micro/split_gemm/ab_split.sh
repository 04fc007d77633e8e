for sp in 1 0; do
  for cfg in c2:f32 c3:f32 c5share:bf16x2 c3:f32x3; do
    wl=${cfg%%:*}; dt=${cfg#*:}
    echo -n "split=$sp $wl $dt: "
    MVSDF_SPLIT_GEMM=$sp python bench.py --workload $wl --dtype $dt --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']['differentiable']; print(round(d['ms_per_step'], 4), 'ms; diff fwd', round(k['ms_forward'], 4), 'bwd', round(k['ms_backward'], 4), 'loss', d['loss'])"
  done
done
