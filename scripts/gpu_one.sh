cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/one; rm -f gpurun_out/one/*
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_spoco.py tests/test_gpu_determinism.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/one/out.txt
cat gpurun_out/one/out.txt
