#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/wgv; mkdir -p $O gpurun_out/r6a; rm -f $O/times.txt
for s in "1 16 16 16 32 32" "2 64 64 64 64 64" "1 24 40 40 64 32"; do HARNESS_CHECK_ARITH=1 WG_ONE=3 build/wg_harness_new $s 3 3 | grep CHECK; HARNESS_CHECK_ARITH=1 WG_ONE=0 build/wg_harness_new $s 3 3 | grep CHECK; HARNESS_CHECK_ARITH=1 WG_ONE=4 build/wg_harness_new $s 3 3 | grep CHECK;  done > $O/check.txt 2>&1
cat $O/check.txt
bash scripts/wg_variants_run.sh base new newu4 newu16 > /dev/null 2>&1
cat $O/times.txt
WG_ONE=3 WG_GZERO=0.5 build/wg_harness_oldtrace 2 128 128 128 32 32 10 3 > $O/trace_old.txt 2>&1
WG_ONE=3 WG_GZERO=0.5 build/wg_harness_newtrace 2 128 128 128 32 32 10 3 > $O/trace_new.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_determinism.py -q -m gpu -k "wgrad or determin" > gpurun_out/r6a/pytest_wgrad.txt 2>&1; tail -3 gpurun_out/r6a/pytest_wgrad.txt
timeout 600 python -m pytest tests/test_gpu_unet.py -q -m gpu -k "bit_identical or benchmark_config" > gpurun_out/r6a/pytest_digest.txt 2>&1; tail -3 gpurun_out/r6a/pytest_digest.txt
timeout 600 python -m pytest tests/test_gpu_train_multi_gpu.py tests/test_gpu_amp_reference.py tests/test_gpu_predict.py -q -s -m gpu > gpurun_out/r6a/pytest_new2.txt 2>&1; tail -3 gpurun_out/r6a/pytest_new2.txt
