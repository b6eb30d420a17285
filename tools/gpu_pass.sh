cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$PWD/llama.cpp.dsp_amd/lib
for i in 1 2 3; do
  for v in "" norc; do
    n=libmi355q${v:+_$v}.so
    echo -n "$n: "; MI355Q_LIB=$L/$n timeout -k 10 200 python tools/loaderonly.py 2>&1 | tail -1
  done
done
