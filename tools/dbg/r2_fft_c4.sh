R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "twenty_object" 2>&1 | grep -E "^E  |passed|failed" | head -20
timeout 600 python bench.py --config fft --steps 20 --warmup 3 > gpurun_out/r2_fft_bench.json 2> gpurun_out/r2_fft_bench.err; tail -c 1800 gpurun_out/r2_fft_bench.json; tail -3 gpurun_out/r2_fft_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2_fft_kt -- python3 $R/bench.py --config fft --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
DB=$(find $R/gpurun_out/r2_fft_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/round2_fft_kernel_stats.txt | head -16
rm -rf $R/gpurun_out/r2_fft_kt
cd $R
timeout 900 python bench.py --config c4 --steps 3 --warmup 1 > gpurun_out/r2_c4_bench.json 2> gpurun_out/r2_c4_bench.err; tail -c 1500 gpurun_out/r2_c4_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2_c4_kt -- python3 $R/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
DB=$(find $R/gpurun_out/r2_c4_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/round2_c4_kernel_stats.txt | head -14
rm -rf $R/gpurun_out/r2_c4_kt
