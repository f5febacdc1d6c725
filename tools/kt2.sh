# kernel trace of frozen epochs with a chosen algorithm: tools/kt2.sh <tag> <algo> <workload>...   -> gpurun_out/<tag>_<w>_trace.txt
set -e
R=$GRAFT_REPO_ROOT
tag=$1; shift
algo=$1; shift
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  O=$R/gpurun_out/kt_${tag}_$w
  rm -rf $O
  rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/run_frozen.py $w 10 $algo $KT_OPTS > /dev/null 2>&1
  (cd $R && python tools/trace_epoch.py $O > $R/gpurun_out/${tag}_${w}_trace.txt)
  rm -rf $O
done
