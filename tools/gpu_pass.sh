cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for pc in "100 128" "250 256" "1000 1024" "2000 2048" "4000 4096"; do
  set -- $pc
  echo "pos $1, n_ctx $2: $(timeout -k 10 120 python tools/loaderonly.py --pos $1 --n-ctx $2 2>&1 | tail -1)"
done
