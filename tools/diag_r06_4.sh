set -u
O=gpurun_out/r06/diag4
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_idr.py tests/test_gpu_trace.py tests/test_gpu_f32x3.py tests/test_gpu_deferred.py tests/test_gpu_native_step.py -q -m gpu -s > $O/pytest.txt 2>&1
grep -v "^$" $O/pytest.txt | grep -i "worst sampled\|passed\|failed\|error\|rays hit\|Error" | tail -60
tail -30 $O/pytest.txt
