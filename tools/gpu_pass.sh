cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_plan
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmc_plan/a -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pp > gpurun_out/pmc_plan_a.log 2>&1; echo "exit $?"
timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM --output-format csv -d gpurun_out/pmc_plan/b -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pp > gpurun_out/pmc_plan_b.log 2>&1; echo "exit $?"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_plan/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_plan" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} n={len(v):3d} mean {sum(v)/len(v):.4g}")
PY
exit 0
