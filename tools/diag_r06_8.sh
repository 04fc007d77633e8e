set -u
O=gpurun_out/r06/diag8
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_options.py tests/test_gpu_diff.py tests/test_gpu_alt_paths.py -q -m gpu > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
