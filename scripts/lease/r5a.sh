# round 5, first GPU call: 16-bit storage tests, digests old vs new library, amp bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_storage16.py -q 2>&1 | tail -40 > $O/storage16.txt
TEM_LIB=$GRAFT_REPO_ROOT/build/base/libtem_hip_r4.so timeout 600 python scripts/digest_default.py > $O/digest_r4.json 2> $O/digest_r4.err
timeout 600 python scripts/digest_default.py > $O/digest_r5.json 2> $O/digest_r5.err
cmp $O/digest_r4.json $O/digest_r5.json > $O/digest_cmp.txt 2>&1 && echo IDENTICAL >> $O/digest_cmp.txt
timeout 600 python bench.py --precision amp --steps 20 --warmup 5 > $O/bench_amp16.txt 2>&1
TEM_AMP_STORAGE=32 timeout 600 python bench.py --precision amp --steps 20 --warmup 5 > $O/bench_amp32.txt 2>&1
timeout 600 python bench.py --precision amp_bf16 --steps 20 --warmup 5 > $O/bench_ampbf16.txt 2>&1
tail -5 $O/storage16.txt; cat $O/digest_cmp.txt; tail -2 $O/bench_amp16.txt | cut -c1-600; tail -1 $O/bench_amp32.txt | cut -c1-300
