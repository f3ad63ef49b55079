# scratch script for one-off gpurun calls: `gpurun -- 'bash scripts/gpu_one.sh'`; default = the GPU suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/one; rm -f gpurun_out/one/*
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/one/out.txt
cat gpurun_out/one/out.txt
