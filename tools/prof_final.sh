# final round-1 evidence: bench lines (P4 with the CPU baseline, S8) and the kernel-trace summary of the P4 bench command
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
python $R/bench.py > $R/gpurun_out/bench_p4.json 2> $R/gpurun_out/bench_p4.err
python $R/bench.py --workload S8 --no-cpu-baseline > $R/gpurun_out/bench_s8.json 2> $R/gpurun_out/bench_s8.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_g.log 2>&1
cut -c1-400 $R/gpurun_out/bench_p4.json; cut -c1-400 $R/gpurun_out/bench_s8.json
find $R/gpurun_out/prof_g -name "*kernel_stats*" | head
