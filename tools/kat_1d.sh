#!/bin/bash
# 1-D BO known-answer scan: how many of 8 seeds reach (0.852733, 2.273928) after 15 / 20 iterations, per maximiser branch
cd ${GRAFT_REPO_ROOT:-.}
for strat in direct multistart; do
  for iters in 15 20; do
    reached=0
    for seed in 1 2 3 4 5 6 7 8; do
      out=$(SLS_GLOBAL_SEARCH=$strat sequential-line-search_amd/bin/bayesian_optimization_1d 1 $iters $seed | tail -1)
      x=$(echo "$out" | awk '{print $4}'); y=$(echo "$out" | awk '{print $6}')
      ok=$(python3 -c "print(int(abs($x-0.852733)<2e-2 and abs($y-2.273928)<2e-2))")
      reached=$((reached+ok))
      echo "  $strat iters=$iters seed=$seed: x=$x y=$y ok=$ok"
    done
    echo "KAT $strat iters=$iters: reached $reached / 8"
  done
done
