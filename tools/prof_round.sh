set -o pipefail
export NST_TEST_REPORT=$PWD/gpurun_out/parity_report.txt; rm -f $NST_TEST_REPORT
python -m pytest tests -m gpu -q > gpurun_out/gputest10.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/gputest10.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/bench_profiled.json 2> $R/gpurun_out/bench_profiled.err; echo "stats rc $?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/pmc_run.py 3 2 > /dev/null 2>&1; echo "fetch rc $?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/pmc_run.py 3 2 > /dev/null 2>&1; echo "write rc $?"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/tools/pmc_run.py 3 2 > /dev/null 2>&1; echo "mfma rc $?"
cd $R
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -2
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; echo "bench rc $?"
