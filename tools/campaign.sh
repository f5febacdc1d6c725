# Round-3 evidence campaign (one gpurun call): PMC traffic + SQ counters, kernel stats of the bench command,
# per-launch epoch traces, bench lines of every workload.  Outputs under gpurun_out/camp (copy into profiles/).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/camp
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in c4 c3 c2 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$w -- python3 $R/tools/run_frozen.py $w 6 filtered > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w_$w -- python3 $R/tools/run_frozen.py $w 6 filtered > /dev/null 2>&1
  (cd $R && python tools/pmc_traffic.py $w $O/f_$w $O/w_$w > /dev/null)
  rm -rf $O/f_$w $O/w_$w
  echo "pmc $w done"
done
for w in c4 c5; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_$w -- python3 $R/tools/run_frozen.py $w 6 filtered > /dev/null 2>&1
  (cd $R && python tools/pmc_summary.py $O/sq_$w > $O/sq_${w}_summary.txt)
  rm -rf $O/sq_$w
done
cd $R
cp profiles/pmc_traffic.json profiles/r03_*_pmc_traffic.txt $O/
echo "bench"
python bench.py --steps 20 --warmup 6 > $O/c4_bench.json 2> $O/c4_bench.err
for w in c3 c2 c5; do python bench.py --workload $w --steps 20 --warmup 6 --cpu-sample 0 > $O/${w}_bench.json 2>/dev/null; done
python bench.py --via ctx --steps 20 --warmup 6 > $O/c4_bench_via_ctx.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 $R/bench.py --steps 10 --warmup 6 --cpu-sample 0 --other-data 0 > $O/stats_c4_bench.json 2>/dev/null
cp $(find $O/stats_c4 -name "*kernel_stats.csv" | head -1) $O/c4_kernel_stats.csv
rm -rf $O/stats_c4
for w in c4 c3 c5 c2; do rocprofv3 --kernel-trace --output-format csv -d $O/kt_$w -- python3 $R/tools/run_frozen.py $w 10 filtered > /dev/null 2>&1; (cd $R && python tools/trace_epoch.py $O/kt_$w > $O/epoch_trace_$w.txt); rm -rf $O/kt_$w; done
(cd $R && python tools/bench_te.py c4 c3 > $O/topographic_error_ms.txt 2>&1)
echo campaign ok
