cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O; rm -f $O/*
timeout 1200 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_predict.py -q -m gpu 2>&1 | tail -15 > $O/pytest_new.txt
timeout 1200 python -m pytest tests/test_gpu_trainer.py -q -m gpu -k "graph" 2>&1 | tail -5 >> $O/pytest_new.txt
bash scripts/gpu_power.sh
cp gpurun_out/power/power.txt $O/power.txt
TEM_BENCH_FORCE_DDP=1 TEM_HIP_GRAPH=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>$O/ddp_graph.err | tail -1 > $O/bench_ddp_graph.json
TEM_BENCH_FORCE_DDP=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_ddp_eager.json
TEM_HIP_GRAPH=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_graph.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_eager.json
