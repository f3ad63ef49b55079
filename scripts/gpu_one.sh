cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/one
free -g | head -2 > gpurun_out/one/mem.txt
timeout 1500 python -m pytest tests/test_gpu_unet.py -q -m gpu -x -s -k "full_size_sample_matches_oracle" 2>&1 | grep -v Warning | tail -12 > gpurun_out/one/out.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "z_sliding_kernels_agree" 2>&1 | tail -3 >> gpurun_out/one/out.txt
