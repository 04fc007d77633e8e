cd $GRAFT_REPO_ROOT
python -m pytest "tests/test_gpu_idr.py" tests/test_gpu_shapes.py -m gpu -x -q 2>&1 | tail -2
DEV=$PWD/mvsdf_amd/libmvsdf_hip_dev.so
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-22s' % '$tag', 'ms %.4f'%d['ms_per_step'], 'sphere %.3f samples %.3f diff %.3f (fwd %.3f bwd %.3f)'%(k['k_sphere_trace']['ms_per_step'], k['k_ray_samples']['ms_per_step'], k['differentiable']['ms_per_step'], k['differentiable']['ms_forward'], k['differentiable']['ms_backward']))"; }
EXTRA="--width 512 --steps 60" run w512-x3 MVSDF_LIB=$DEV
EXTRA="--width 512 --steps 60" run w512-f32chain MVSDF_LIB=$DEV MVSDF_CHAIN_X3=0
EXTRA="--workload shipped --steps 10 --warmup 2" run shipped-x3 MVSDF_LIB=$DEV
EXTRA="--workload shipped --steps 10 --warmup 2" run shipped-f32chain MVSDF_LIB=$DEV MVSDF_CHAIN_X3=0
