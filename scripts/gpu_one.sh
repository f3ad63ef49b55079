cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/one
timeout 900 python -m pytest tests/test_gpu_unet.py -q -m gpu -x -k "weight_gradient_arithmetic" 2>&1 | grep -v Warning | tail -60 > gpurun_out/one/out.txt
