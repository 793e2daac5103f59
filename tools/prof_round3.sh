# Round-3 profiles (run on the GPU box from the repo root): kernel stats of the bench command, then separate --pmc passes
# (HBM FETCH_SIZE / WRITE_SIZE, MFMA busy) over a few single-stream L=2 closures, summarised into gpurun_out/r03_*.
set -o pipefail
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/r03_bench_profiled.json 2> $R/gpurun_out/bench_profiled.err; echo "stats rc $?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/pmc_run.py 3 2 > /dev/null 2>&1; echo "fetch rc $?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/pmc_run.py 3 2 > /dev/null 2>&1; echo "write rc $?"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/tools/pmc_run.py 3 2 > /dev/null 2>&1; echo "mfma rc $?"
cd $R
cp $(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r03_bench_default_kernel_stats.csv
python tools/summarize_pmc.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r03_pmc_hbm_traffic_per_closure.txt gpurun_out/r03_pmc_hbm_traffic.json; echo "traffic rc $?"
python tools/summarize_pmc.py mfma gpurun_out/pmc_mfma gpurun_out/r03_pmc_mfma_util_conv.txt; echo "mfma summary rc $?"
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"
