set -u
O=gpurun_out/r06/diag6
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_alt_paths.py tests/test_gpu_f32x3.py tests/test_gpu_fullsize.py tests/test_gpu_trace.py tests/test_gpu_bf16s.py tests/test_gpu_lazy.py -q -m gpu > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_pp_$i.json 2> $O/err.txt; done
python3 bench.py --no-cpu-baseline > $O/long_pp.json 2>$O/err.txt
python3 bench.py --no-cpu-baseline --workload c5share --dtype bf16x2 > $O/long_c5_pp.json 2>$O/err.txt
python3 bench.py --no-cpu-baseline --workload c3 --steps 60 > $O/long_c3_pp.json 2>$O/err.txt
for f in drv_pp_1 drv_pp_2 long_pp long_c5_pp long_c3_pp; do python3 -c "
import json;d=json.load(open('$O/$f.json'));t=d['timing'];k=d['roofline']['kernels']
print('$f', 'ms %.3f' % d['ms_per_step'], 'kernel_ms', t['kernel_ms_per_step'], 'sphere %.3f samples %.3f' % (k['k_sphere_trace']['ms_per_step'], k['k_ray_samples']['ms_per_step']), 'diff', k['differentiable']['ms_per_step'], 'roofline', d['roofline']['kernel'][:16], d['roofline']['frac'])"; done
