cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-22s' % '$tag', 'ms %.4f'%d['ms_per_step'], 'sphere %.3f samples %.3f diff %.3f (fwd %.3f bwd %.3f)'%(k['k_sphere_trace']['ms_per_step'], k['k_ray_samples']['ms_per_step'], k['differentiable']['ms_per_step'], k['differentiable']['ms_forward'], k['differentiable']['ms_backward']))"; }
for rep in 1 2; do
EXTRA="--workload c3" run c3-default A=1
EXTRA="--workload c3" run c3-split0 MVSDF_SPLIT_ROWS=0
EXTRA="--workload c3" run c3-split1 MVSDF_SPLIT_ROWS=1
EXTRA="--workload c5share --dtype bf16x2" run c5s-default A=1
EXTRA="--workload c5share --dtype bf16x2" run c5s-split0 MVSDF_SPLIT_ROWS=0
EXTRA="--workload c5share --dtype bf16x2" run c5s-split1 MVSDF_SPLIT_ROWS=1
done
EXTRA="--workload shipped --steps 10 --warmup 2" run shipped-default A=1
EXTRA="--workload shipped --steps 10 --warmup 2" run shipped-split0 MVSDF_SPLIT_ROWS=0
EXTRA="--workload shipped --steps 10 --warmup 2" run shipped-split1 MVSDF_SPLIT_ROWS=1
