# Evidence campaign of a round (two gpurun calls): tools/campaign.sh pmc | bench   -> gpurun_out/camp (copy into profiles/)
#   pmc   : HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the four workloads -> profiles/pmc_traffic.json entries
#           of THIS build, SQ counters of c4 and c5
#   bench : the driver's bench line (with other_workloads and fit), per-workload lines, --via ctx, rocprofv3 kernel stats of
#           the bench command, per-launch epoch traces, topographic-error timing, a rank's share
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/camp
mkdir -p $O
TAG=${CAMPAIGN_TAG:-r04}
cd /tmp && export TMPDIR=/tmp
if [ "$1" = "pmc" ]; then
  for w in c4 c3 c2 c5; do
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$w -- python3 $R/tools/run_frozen.py $w 6 filtered > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w_$w -- python3 $R/tools/run_frozen.py $w 6 filtered > /dev/null 2>&1
    (cd $R && CAMPAIGN_TAG=$TAG python tools/pmc_traffic.py $w $O/f_$w $O/w_$w > /dev/null)
    rm -rf $O/f_$w $O/w_$w
    echo "pmc $w done"
  done
  for w in c4 c5; do
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_$w -- python3 $R/tools/run_frozen.py $w 6 filtered > /dev/null 2>&1
    (cd $R && python tools/pmc_summary.py $O/sq_$w > $O/${TAG}_${w}_pmc_sq_summary.txt)
    rm -rf $O/sq_$w
    echo "sq $w done"
  done
  cp $R/profiles/pmc_traffic.json $O/
  cp $R/profiles/${TAG}_*_pmc_traffic.txt $O/ 2>/dev/null || true
  echo campaign pmc ok
  exit 0
fi
cd $R
python bench.py --steps 20 --warmup 5 > $O/${TAG}_final_c4_bench.json 2> $O/c4_bench.err
echo "bench c4 done"
for w in c3 c2 c5; do python bench.py --workload $w --steps 20 --warmup 5 --cpu-sample 0 > $O/${TAG}_final_${w}_bench.json 2>/dev/null; done
python bench.py --via ctx --steps 20 --warmup 5 > $O/${TAG}_final_c4_bench_via_ctx.json 2>/dev/null
echo "bench others done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 $R/bench.py --steps 10 --warmup 5 --cpu-sample 0 --other-data 0 --other-workloads 0 --fit 0 > $O/stats_c4_bench.json 2>/dev/null
cp $(find $O/stats_c4 -name "*kernel_stats.csv" | head -1) $O/${TAG}_final_c4_kernel_stats.csv
rm -rf $O/stats_c4
for w in c4 c3 c5 c2; do rocprofv3 --kernel-trace --output-format csv -d $O/kt_$w -- python3 $R/tools/run_frozen.py $w 10 filtered > /dev/null 2>&1; (cd $R && python tools/trace_epoch.py $O/kt_$w > $O/${TAG}_final_${w}_epoch_trace.txt); rm -rf $O/kt_$w; done
echo "traces done"
cd $R
python tools/bench_te.py c4 c3 > $O/${TAG}_final_topographic_error_ms.txt 2>&1
(for a in "c4 rows=500000 0" "c4 rows=250000 0" "c4 rows=125000 0" "c4 rows=125000 ranks=8 0" "c2 0" "c5 ranks=8 0"; do python tools/bench_rank_shard.py $a 2>/dev/null; done) > $O/${TAG}_final_rank_share.txt
echo campaign bench ok
