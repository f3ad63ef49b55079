cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/one; rm -f gpurun_out/one/*
python scripts/nan_probe.py > gpurun_out/one/out.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "absmax or wgrad_fp16_two" 2>&1 | tail -3 >> gpurun_out/one/out.txt
