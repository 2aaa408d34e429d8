set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_f.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_f1 -o p -- python $R/tools/run_back_fwd.py > $R/gpurun_out/pmc_f1.log 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f2 -o p -- python $R/tools/run_back_fwd.py > $R/gpurun_out/pmc_f2.log 2>&1 || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_f3 -o p -- python $R/tools/run_back_fwd.py > $R/gpurun_out/pmc_f3.log 2>&1 || true
tail -1 $R/gpurun_out/bench_f.log | cut -c1-300
ls $R/gpurun_out/pmc_f1 $R/gpurun_out/pmc_f2 | head
