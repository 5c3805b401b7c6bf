# Counter passes for tools/pmc_traffic.py: --pmc only with --kernel-trace (no other trace domain), one counter per pass.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc; rm -rf $O; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pn_fetch -- python3 $R/tools/run_forward.py phasenet 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pn_write -- python3 $R/tools/run_forward.py phasenet 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/eqt_fetch -- python3 $R/tools/run_forward.py eqtransformer 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/eqt_write -- python3 $R/tools/run_forward.py eqtransformer 6 > /dev/null 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python tools/pmc_traffic.py gpurun_out/pmc ${TRAFFIC_JSON:-r04_traffic.json} | tail -16
find $O -name "*counter_collection.csv" | head
