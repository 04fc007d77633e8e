# correctness (bwd_ab vs the float64 oracle, the reference fixtures) and the step with / without the three-term weight-gradient kernel on one box
cd $GRAFT_REPO_ROOT
timeout 300 python tools/micro/chain_x3/bwd_ab.py 2>&1 | grep " x3 " | cut -c1-230
python -m pytest tests/test_gpu_diff.py tests/test_gpu_idr.py tests/test_gpu_native_step.py tests/test_gpu_options.py -m gpu -x -q 2>&1 | tail -3
DEV=$PWD/mvsdf_amd/libmvsdf_hip_dev.so
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-22s' % '$tag', 'ms %.4f'%d['ms_per_step'], 'sphere %.3f samples %.3f diff %.3f (fwd %.3f bwd %.3f)'%(k['k_sphere_trace']['ms_per_step'], k['k_ray_samples']['ms_per_step'], k['differentiable']['ms_per_step'], k['differentiable']['ms_forward'], k['differentiable']['ms_backward']))"; }
for rep in 1 2; do
run c2-wx3 MVSDF_LIB=$DEV
run c2-wf32 MVSDF_LIB=$DEV MVSDF_WGRAD_X3=0
done
EXTRA="--workload c3" run c3-wx3 MVSDF_LIB=$DEV
EXTRA="--workload c3" run c3-wf32 MVSDF_LIB=$DEV MVSDF_WGRAD_X3=0
EXTRA="--workload c5share --dtype bf16x2" run c5s-wx3 MVSDF_LIB=$DEV
EXTRA="--workload c5share --dtype bf16x2" run c5s-wf32 MVSDF_LIB=$DEV MVSDF_WGRAD_X3=0
EXTRA="--workload shipped --steps 10 --warmup 2" run shipped-wx3 MVSDF_LIB=$DEV
EXTRA="--workload shipped --steps 10 --warmup 2" run shipped-wf32 MVSDF_LIB=$DEV MVSDF_WGRAD_X3=0
