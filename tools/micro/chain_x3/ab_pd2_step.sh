cd $GRAFT_REPO_ROOT
DEV=$PWD/mvsdf_amd/libmvsdf_hip_dev.so
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-22s' % '$tag', 'ms %.4f'%d['ms_per_step'], 'sphere %.3f samples %.3f diff %.3f (fwd %.3f bwd %.3f)'%(k['k_sphere_trace']['ms_per_step'], k['k_ray_samples']['ms_per_step'], k['differentiable']['ms_per_step'], k['differentiable']['ms_forward'], k['differentiable']['ms_backward']))"; }
for rep in 1 2 3; do
EXTRA="--workload c3" run c3-pd0 MVSDF_LIB=$DEV
EXTRA="--workload c3" run c3-pd2 MVSDF_LIB=$PWD/mvsdf_amd/libmvsdf_hip_devpd2.so
EXTRA="--workload c3" run c3-pd4 MVSDF_LIB=$PWD/mvsdf_amd/libmvsdf_hip_devpd4.so
done
for rep in 1 2; do
EXTRA="--workload c5share --dtype bf16x2" run c5s-pd0 MVSDF_LIB=$DEV
EXTRA="--workload c5share --dtype bf16x2" run c5s-pd4 MVSDF_LIB=$PWD/mvsdf_amd/libmvsdf_hip_devpd4.so
EXTRA="--scaling strong --steps 60" run strong-pd0 MVSDF_LIB=$DEV
EXTRA="--scaling strong --steps 60" run strong-pd4 MVSDF_LIB=$PWD/mvsdf_amd/libmvsdf_hip_devpd4.so
done
