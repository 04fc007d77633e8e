cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$tag', 'ms %.4f'%d['ms_per_step'], 'sphere %.3f samples %.3f diff %.3f tailrows %d'%(k['k_sphere_trace']['ms_per_step'], k['k_ray_samples']['ms_per_step'], k['differentiable']['ms_per_step'], k['k_sphere_trace']['rows_min_sdf_tail']))"; }
run base A=1
run split MVSDF_SPLIT_ROWS=1
run tail2 MVSDF_TAIL=2
run both MVSDF_SPLIT_ROWS=1 MVSDF_TAIL=2
run base2 A=1
EXTRA="--workload c5share" run c5share A=1
EXTRA="--workload c3" run c3 A=1
EXTRA="--width 512 --steps 60" run w512 A=1
