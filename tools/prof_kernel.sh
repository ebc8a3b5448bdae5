# GPU box: kernel trace + PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) of one probe script, summarised for
# the kernels whose name contains FILTER.  usage: prof_kernel.sh TAG FILTER PROBE.py [ENV=VAL ...]
set -e
TAG=$1; FILTER=$2; PROBE=$3; shift 3
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
P="python3 $R/$PROBE"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- $P > $O/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -- $P > $O/pmc1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_lds -- $P > $O/pmc2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- $P > $O/pmc3.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- $P > $O/pmc4.log 2>&1
python3 $R/tools/pmc_summary.py $O/summary.json "$FILTER" $O/trace $O/pmc_mfma $O/pmc_lds $O/pmc_fetch $O/pmc_write
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete
