cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/suite; mkdir -p $O; rm -f $O/*
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/pytest_gpu.txt
