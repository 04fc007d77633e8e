set -u
O=gpurun_out/r06/diag2
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_deferred.py tests/test_gpu_native_step.py -x -q -m gpu > $O/pytest_deferred.txt 2>&1
tail -30 $O/pytest_deferred.txt
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_$i.json 2> $O/drv_$i.err; tail -3 $O/drv_$i.err; done
python3 bench.py --no-cpu-baseline > $O/long.json 2>$O/long.err
MVSDF_DEFERRED_STEP=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_classic.json 2> $O/drv_classic.err
python3 tools/host_sections.py > $O/host_sections.txt 2>&1
cat $O/host_sections.txt
for f in drv_1 drv_2 long drv_classic; do python3 -c "import json;d=json.load(open('$O/$f.json'));print('$f', d['ms_per_step'], d['value'])"; done
