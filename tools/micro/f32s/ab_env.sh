# usage: bash tools/micro/f32s/ab_env.sh "<ENV=VAL ...>" ... : bench.py --dtype f32x3 at c2 and c3 under each environment
for envs in "$@"; do
  for wl in ${WLS:-c2 c3}; do
    echo -n "[$envs] $wl: "
    env $envs python bench.py --workload $wl --dtype f32x3 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 4), 'ms', {k: round(v['ms'], 4) if isinstance(v, dict) and 'ms' in v else None for k, v in d['roofline'].get('kernels', {}).items()})"
  done
done
