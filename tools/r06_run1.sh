mkdir -p gpurun_out/r06
cd tools/probes
for CFG in "0 0 2 0" "1 0 3 2" "1 0 2 0" "0 0 3 2"; do set -- $CFG
for BW in 0 16 24 32 48 64; do for BN in 1 2 3; do
  [ $BW = 0 ] && [ $BN != 1 ] && continue
  r=$(SLS_POTRF_BAND_W=$BW SLS_POTRF_BAND=$BN SLS_POTRF_STREAM=$1 SLS_POTRF_SPLIT=$2 SLS_POTRF_DNBO=$3 SLS_POTRF_DNEAR=$4 POTRF_BENCH_QUICK=1 POTRF_BENCH_TRACE=1 timeout 100 ./bin/potrf_bench 8192 2>&1 | grep -E "dataflow single|chain waited" | tr '\n' ' ' | sed 's/info=.*max|L/max|L/' )
  echo "stream=$1 split=$2 nbo=$3 near=$4 bandw=$BW band=$BN : $r"
done; done; done > ../../gpurun_out/r06/potrf8192_band_scan.log 2>&1
cat ../../gpurun_out/r06/potrf8192_band_scan.log | cut -c1-210
