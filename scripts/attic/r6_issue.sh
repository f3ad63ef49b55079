cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x32b; timeout 120 build/issue_bench_f32 > gpurun_out/x32b/issue_f32.txt 2>&1
