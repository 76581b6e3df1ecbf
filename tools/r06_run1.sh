mkdir -p gpurun_out/r06
cd tools/probes
for N in 4096 3072; do
for H in -1 0 2; do
for KM in 8 16 32; do for RM in 8 16 24 99; do for W1 in 0 64 128; do
  [ $H = -1 ] && { [ $KM != 8 ] || [ $RM != 8 ]; } && continue
  if [ $W1 = 0 ]; then unset SLS_POTRI_W1; else export SLS_POTRI_W1=$W1; fi
  r=$(SLS_POTRI_HYBRID=$H SLS_POTRI_HYB_KMAX=$KM SLS_POTRI_HYB_RMIN=$RM POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 timeout 100 ./bin/potrf_bench $N 2>&1 | grep -E "potri fused single|K\^-1" | head -2 | tr '\n' ' ' | sed 's/info=0 abort=0 applicable=1//; s/(max |value|.*//')
  echo "N=$N H=$H kmax=$KM rmin=$RM W1=$W1 : $r"
done; done; done; done; done > ../../gpurun_out/r06/potri_hybrid2_scan.log 2>&1
sort -t: -k2 ../../gpurun_out/r06/potri_hybrid2_scan.log | grep "N=4096" | sort -k9 -n | head -12 | cut -c1-160
grep "N=3072" ../../gpurun_out/r06/potri_hybrid2_scan.log | sort -k9 -n | head -6 | cut -c1-160
