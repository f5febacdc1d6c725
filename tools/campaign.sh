set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/camp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in c4 c3 c2 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$w -- python3 $R/tools/run_frozen.py $w 5 filtered > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w_$w -- python3 $R/tools/run_frozen.py $w 5 filtered > /dev/null 2>&1
  (cd $R && python tools/pmc_traffic.py $w $O/f_$w $O/w_$w > /dev/null)
  echo "pmc $w done"
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_c4 -- python3 $R/tools/run_frozen.py c4 4 filtered > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_c3 -- python3 $R/tools/run_frozen.py c3 4 filtered > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/sq_c4 > $O/sq_c4_summary.txt
python tools/pmc_summary.py $O/sq_c3 > $O/sq_c3_summary.txt
cp profiles/pmc_traffic.json profiles/r02_*_pmc_traffic.txt $O/
echo "bench"
python bench.py --steps 20 --warmup 4 > $O/c4_bench.json 2> $O/c4_bench.err
for w in c3 c2 c5; do python bench.py --workload $w --steps 20 --warmup 4 --cpu-sample 0 > $O/${w}_bench.json 2>/dev/null; done
python bench.py --via ctx --steps 20 --warmup 4 > $O/c4_bench_via_ctx.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 --other-data 0 > $O/stats_c4_bench.json 2>/dev/null
for w in c4 c2; do rocprofv3 --kernel-trace --output-format csv -d $O/kt_$w -- python3 $R/tools/run_frozen.py $w 8 filtered > /dev/null 2>&1; (cd $R && python tools/trace_epoch.py $O/kt_$w > $O/epoch_trace_$w.txt); done
echo campaign ok
