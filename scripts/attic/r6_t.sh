cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/x32a
timeout 900 python -m pytest tests/test_gpu_unet.py -q -s -k "exact_fp32" 2>&1 | tail -12 > gpurun_out/x32a/pytest2.txt; cat gpurun_out/x32a/pytest2.txt
