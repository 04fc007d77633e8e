set -u
O=gpurun_out/r06/diag5
mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu > $O/pytest_all.txt 2>&1
tail -15 $O/pytest_all.txt
