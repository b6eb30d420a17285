cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export MI355Q_LIB=$PWD/llama.cpp.dsp_amd/lib/libmi355q_dbg.so
timeout -k 10 300 python tools/planstamps.py --layers 32 --csv gpurun_out/stamps32.csv > gpurun_out/stamps32.txt 2>&1; echo rc=$?
wc -l gpurun_out/stamps32.csv
