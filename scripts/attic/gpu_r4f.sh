cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O; rm -f $O/*
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > $O/pytest_gpu.txt
