set -u
O=gpurun_out/r06/diag9
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_f32x3.py tests/test_gpu_fullsize.py tests/test_gpu_trace.py tests/test_gpu_bf16s.py tests/test_gpu_bf16.py tests/test_gpu_shapes.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv.json 2> $O/err.txt
python3 bench.py --no-cpu-baseline > $O/long.json 2>$O/err.txt
python3 bench.py --no-cpu-baseline --workload c5share --dtype bf16x2 > $O/long_c5.json 2>$O/err.txt
python3 bench.py --no-cpu-baseline --workload c5share > $O/long_c5_f32x3.json 2>$O/err.txt
python3 bench.py --no-cpu-baseline --dtype bf16x2 > $O/long_c2_bf16x2.json 2>$O/err.txt
python3 bench.py --no-cpu-baseline --workload shipped --steps 20 --warmup 5 > $O/shipped.json 2>$O/err.txt
for f in drv long long_c5 long_c5_f32x3 long_c2_bf16x2 shipped; do python3 -c "
import json;d=json.load(open('$O/$f.json'));t=d['timing'];k=d['roofline']['kernels']
print('$f', 'ms %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % t['kernel_ms_per_step'], 'sphere %.3f samples %.3f' % (k['k_sphere_trace']['ms_per_step'], k['k_ray_samples']['ms_per_step']), 'diff %.3f' % k['differentiable']['ms_per_step'], d['roofline']['kernel'][:16], '%.3f' % d['roofline']['frac'])"; done
