cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
rm -rf gpurun_out/pmc_r3a gpurun_out/pmc_r3b gpurun_out/pmc_r3c
cd /tmp
timeout -k 10 280 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_r3a -- python3 $R/tools/loaderonly.py > $R/gpurun_out/pmc_r3a.log 2>&1; echo rc=$?
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_r3b -- python3 $R/tools/loaderonly.py > $R/gpurun_out/pmc_r3b.log 2>&1; echo rc=$?
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_r3c -- python3 $R/tools/loaderonly.py > $R/gpurun_out/pmc_r3c.log 2>&1; echo rc=$?
cd $R
python3 tools/summarize_pmc.py gpurun_out/round3_k_plan_counters_raw.md "round 3: k_plan counters (tools/loaderonly.py, position 100, window 128)" gpurun_out/pmc_r3a gpurun_out/pmc_r3b gpurun_out/pmc_r3c | tail -6 | cut -c1-600
