cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_diff.py tests/test_gpu_idr.py tests/test_gpu_native_step.py tests/test_gpu_options.py -m gpu -x -q 2>&1 | tail -4
DEV=$PWD/mvsdf_amd/libmvsdf_hip_dev.so
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-22s' % '$tag', 'ms %.4f'%d['ms_per_step'], 'sphere %.3f samples %.3f diff %.3f (fwd %.3f bwd %.3f)'%(k['k_sphere_trace']['ms_per_step'], k['k_ray_samples']['ms_per_step'], k['differentiable']['ms_per_step'], k['differentiable']['ms_forward'], k['differentiable']['ms_backward']))"; }
run c2-x3 MVSDF_LIB=$DEV
run c2-f32chain MVSDF_LIB=$DEV MVSDF_CHAIN_X3=0
run c2-x3 MVSDF_LIB=$DEV
EXTRA="--workload c5share --dtype bf16x2" run c5share-bf16x2-x3 MVSDF_LIB=$DEV
EXTRA="--workload c5share --dtype bf16x2" run c5share-bf16x2-f32 MVSDF_LIB=$DEV MVSDF_CHAIN_X3=0
EXTRA="--workload c3" run c3-x3 MVSDF_LIB=$DEV
EXTRA="--workload c3" run c3-f32chain MVSDF_LIB=$DEV MVSDF_CHAIN_X3=0
EXTRA="--workload shipped --steps 10 --warmup 2" run shipped-x3 MVSDF_LIB=$DEV
