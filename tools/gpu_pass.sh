cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
MI355Q_LIB=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libmi355q_dbg.so timeout -k 10 300 python tools/planstamps.py --layers 2 --pos 20 > gpurun_out/planstamps.txt 2>&1; echo "exit $?"
grep "^stage" gpurun_out/planstamps.txt | head -8 | cut -c1-400
exit 0
