cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_storage16.py -q -s 2>&1 | tail -60 > $O/storage16.txt
grep -E "passed|failed|16-bit vs" $O/storage16.txt | tail -8
timeout 3000 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_storage16.py 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
