cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/one
TEM_PRECISION=fp32 timeout 900 python -m pytest tests/test_gpu_unet.py -q -m gpu -x -k "mfma_sizes and GroupNorm" 2>&1 | grep -v Warning | tail -40 > gpurun_out/one/out.txt
TEM_OPT_WGRAD_ZS=1 TEM_PRECISION=fp32 timeout 900 python -m pytest tests/test_gpu_unet.py -q -m gpu -x -k "mfma_sizes and GroupNorm" 2>&1 | grep -v Warning | tail -40 > gpurun_out/one/out_old.txt
