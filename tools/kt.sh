# kernel trace of frozen epochs: tools/kt.sh <tag> <workload>...   -> gpurun_out/<tag>_<w>_trace.txt
set -e
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  O=$R/gpurun_out/kt_${tag}_$w
  rm -rf $O
  rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/run_frozen.py $w ${KT_STEPS:-8} filtered $KT_OPTS > /dev/null 2>&1
  (cd $R && python tools/trace_epoch.py $O > $R/gpurun_out/${tag}_${w}_trace.txt)
  rm -rf $O
done
