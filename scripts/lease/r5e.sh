cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_trainer.py -q -x -k "graph_is_dropped or bfloat16" 2>&1 | tail -60 > $O/t.txt
grep -E "^E|passed|failed" $O/t.txt | cut -c1-400 | head -30
timeout 600 python -m pytest tests/test_gpu_storage16.py -q 2>&1 | tail -3
