# usage: bash tools/micro/f32s/ab_dtypes.sh "<workload>:<dtype>" ... : bench.py lines (30 steps)
for cfg in "$@"; do
  wl=${cfg%%:*}; dt=${cfg#*:}
  echo -n "$wl $dt: "
  python bench.py --workload $wl --dtype $dt --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 4), 'ms; sphere', round(d['roofline']['kernels']['k_sphere_trace']['ms_per_step'], 4), 'samples', round(d['roofline']['kernels']['k_ray_samples']['ms_per_step'], 4))"
done
