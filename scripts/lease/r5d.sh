cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_storage16.py -q -s 2>&1 | grep -E "passed|failed|16-bit vs|FAILED" > $O/storage16.txt
cat $O/storage16.txt
timeout 900 python -m pytest tests/test_gpu_trainer.py -q -s -k "mixed_precision or bfloat16 or autocast" 2>&1 | grep -E "loss|passed|failed|FAILED|^E " | cut -c1-500 > $O/trainer_amp.txt
cat $O/trainer_amp.txt
timeout 3000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_storage16.py --deselect tests/test_gpu_trainer.py::test_bfloat16_trainer_against_reference_autocast_run 2>&1 | tail -15 > $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
