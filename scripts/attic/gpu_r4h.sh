cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O; rm -f $O/*
timeout 600 python scripts/fp32_conv_error.py > $O/fp32_conv_error.txt 2>&1
timeout 2400 bash scripts/depth4_bisect.sh > $O/depth4_bisect.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_unet.py -q -m gpu 2>&1 | tail -5 > $O/pytest_unet.txt
