#!/bin/bash
# rocprofv3 evidence for profiles/r03_*: kernel-trace stats of the default bench line shape (active-set compaction on) and
# PMC passes (separate runs, as the MI355X guide prescribes) of the full 65 536-candidate launch shape (SLS_COMPACT=0, so
# that every acq_gemm dispatch in the pass has the same shape and the per-launch means are meaningful).
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --n-local 12 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
export SLS_COMPACT=0
BENCH="python $R/bench.py --steps 1 --warmup 1 --n-local 6 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d $OUT/pmc_mfma -o pmc -- $BENCH > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -40
du -sh $OUT
python $R/tools/summarize_prof.py $OUT $R/gpurun_out/r03/r03 > $OUT/summary.log 2>&1; tail -30 $OUT/summary.log
cd $R && python tools/make_pmc_summary.py gpurun_out/r03/r03 >> $OUT/summary.log 2>&1; tail -3 $OUT/summary.log
