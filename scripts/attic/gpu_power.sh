cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/power; mkdir -p $O; rm -f $O/*
# forward fp16x3 (mode 4) and dgrad bf16x3 (mode 2) on the z-reuse kernel (variant 2 = forced), 32 -> 32 at 2 x 128^3
scripts/kernel_power.sh zr_f16x3_fwd timeout 60 build/pp_harness_r4 2 128 128 128 32 32 4 2 1500 1 0 >> $O/power.txt 2>&1
scripts/kernel_power.sh zr_bf16x3_dgrad timeout 60 build/pp_harness_r4 2 128 128 128 32 32 2 2 1500 0 1 >> $O/power.txt 2>&1
WG_ONE=0 scripts/kernel_power.sh wgrad_zs_bf16x3 timeout 60 build/wg_harness_r4 2 128 128 128 32 32 1500 1 >> $O/power.txt 2>&1
WG_ONE=3 scripts/kernel_power.sh wgrad_zs_f16x2 timeout 60 build/wg_harness_r4 2 128 128 128 32 32 1500 1 >> $O/power.txt 2>&1
WG_ONE=0 scripts/kernel_power.sh wgrad_tr_bf16x3 timeout 60 build/wg_harness_r4 2 128 128 128 32 32 1500 3 >> $O/power.txt 2>&1
WG_ONE=3 scripts/kernel_power.sh wgrad_tr_f16x2 timeout 60 build/wg_harness_r4 2 128 128 128 32 32 1500 3 >> $O/power.txt 2>&1
WG_ONE=1 scripts/kernel_power.sh wgrad_tr_f16 timeout 60 build/wg_harness_r4 2 128 128 128 32 32 1500 3 >> $O/power.txt 2>&1
