set -u
O=gpurun_out/r06/diag1
mkdir -p $O
lscpu | head -30 > $O/lscpu.txt
nproc >> $O/lscpu.txt
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_$i.json 2> $O/drv_$i.err; done
python3 bench.py --no-cpu-baseline > $O/long.json 2>$O/long.err
python3 tools/host_sections.py > $O/host_sections.txt 2>&1
python3 tools/host_native_detail.py > $O/host_native_detail.txt 2>&1
cat $O/host_sections.txt $O/host_native_detail.txt
for i in 1 2 3; do python3 -c "import json;d=json.load(open('$O/drv_$i.json'));print(d['ms_per_step'])"; done
python3 -c "import json;d=json.load(open('$O/long.json'));print(d['ms_per_step'])"
