# PMC passes over frozen epochs: tools/pmc.sh <tag> <workload>  -> gpurun_out/<tag>_<w>_pmc.txt
set -e
R=$GRAFT_REPO_ROOT
tag=$1; w=$2
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_${tag}_$w
rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/tools/run_frozen.py $w 4 filtered > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/tools/run_frozen.py $w 4 filtered > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/run_frozen.py $w 4 filtered > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES TCC_HIT_sum --kernel-trace --output-format csv -d $O/lds -- python3 $R/tools/run_frozen.py $w 4 filtered > /dev/null 2>&1 || true
cd $R
python tools/pmc_summary.py $O/f $O/w $O/sq $O/lds > gpurun_out/${tag}_${w}_pmc.txt
rm -rf $O
