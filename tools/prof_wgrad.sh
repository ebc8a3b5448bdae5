set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_wgpmc_$1
rm -rf $O; mkdir -p $O
export LAYER=$1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 $R/tools/wgrad_probe.py > $O/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -- python3 $R/tools/wgrad_probe.py > $O/pmc1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_lds -- python3 $R/tools/wgrad_probe.py > $O/pmc2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 $R/tools/wgrad_probe.py > $O/pmc3.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python3 $R/tools/wgrad_probe.py > $O/pmc4.log 2>&1
python3 $R/tools/pmc_summary.py $O/summary.json conv5_wgrad $O/trace $O/pmc_mfma $O/pmc_lds $O/pmc_fetch $O/pmc_write
# keep only the small files
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete
