mkdir -p gpurun_out/r06
cd tools/probes
for N in 4096 3072 2048; do
for H in -1 0 1 2 4 8; do
for W1 in 0 32 48 64; do
  if [ $W1 = 0 ]; then unset SLS_POTRI_W1; else export SLS_POTRI_W1=$W1; fi
  r=$(SLS_POTRI_HYBRID=$H POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 timeout 100 ./bin/potrf_bench $N 2>&1 | grep -E "potri fused single|K\^-1" | head -2 | tr '\n' ' ')
  echo "N=$N H=$H W1=$W1 : $r"
done; done; done > ../../gpurun_out/r06/potri_hybrid_scan.log 2>&1
cat ../../gpurun_out/r06/potri_hybrid_scan.log | cut -c1-200
