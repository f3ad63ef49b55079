cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_storage16.py -q 2>&1 | tail -150 > $O/storage16.txt
grep -E "passed|failed" $O/storage16.txt | tail -3
