cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/one; rm -f gpurun_out/one/*
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "out_conv_backward or output_amax" 2>&1 | grep -v Warning | tail -12 > gpurun_out/one/out.txt
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_trainer.py -q -m gpu -x 2>&1 | tail -4 >> gpurun_out/one/out.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/one/out.txt 2>&1
for z in 1 0 1 0; do
TEM_FUSE_OUT_BWD=$z timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse_out_bwd=$z', d['ms_per_step'])" >> gpurun_out/one/out.txt
done
