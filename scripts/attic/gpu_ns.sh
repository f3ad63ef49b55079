#!/bin/bash
# Runs ON THE GPU BOX: k_norm_sums_from_wgrad with parts ablated (TEM_NS_ABL builds), its time inside the real step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ns; mkdir -p $O; rm -rf $O/*
for v in full; do
  lib=$GRAFT_REPO_ROOT/build/var/libtem_hip_$v.so; [ $v = full ] && lib=$GRAFT_REPO_ROOT/torch_em_amd/lib/libtem_hip.so
  (cd /tmp && TEM_LIB=$lib timeout 600 rocprofv3 --kernel-trace -d $O/rp_$v -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/rp_$v.log 2>&1)
  f=$(find $O/rp_$v -name "*_kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/step_trace.py $f | grep -E "k_norm_sums_from_wgrad|k_reduce_slabs_wsum" | grep -v " x " | awk -v v=$v '{print v, $0}' >> $O/ns.txt
done
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
cat $O/ns.txt
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "sums or norm" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_unet.py -q -m gpu -x 2>&1 | tail -2
