# Measurement bundle of a round (tools/measure_bundle.sh r03) (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the same
# command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs, no tracing domains besides the kernel trace).
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err   # the driver's command
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --quick > $O/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -- python3 $R/bench.py --no-cpu-baseline > $O/bench_trace_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fista -- python3 $R/tools/run_config.py --alg fista --iters 5 > $O/fista.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sirt -- python3 $R/tools/run_config.py --alg sirt --iters 10 > $O/sirt.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$grp -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/pmc_$grp.log 2>&1; done
for grp in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $grp --output-format csv -d $O/pmcs_$grp -- python3 $R/tools/run_config.py --alg sirt --iters 1 > $O/pmcs_$grp.log 2>&1; done
for grp in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $grp --output-format csv -d $O/pmcf_$grp -- python3 $R/tools/run_config.py --alg fista --iters 1 > $O/pmcf_$grp.log 2>&1; done
cd $R
python3 tools/prof_summary.py $O/trace $O/kernel_stats.txt > /dev/null
python3 tools/prof_summary.py $O/trace_full $O/kernel_stats_full.txt > /dev/null
python3 tools/prof_summary.py $O/trace_fista $O/kernel_stats_fista.txt > /dev/null
python3 tools/prof_summary.py $O/trace_sirt $O/kernel_stats_sirt.txt > /dev/null
mkdir -p $O/pmc_bench $O/pmc_sirt $O/pmc_fista; mv $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_bench/; mv $O/pmcs_FETCH_SIZE $O/pmcs_WRITE_SIZE $O/pmc_sirt/; mv $O/pmcf_FETCH_SIZE $O/pmcf_WRITE_SIZE $O/pmc_fista/
python3 tools/pmc_summary.py $O/pmc_bench $O/pmc_traffic.txt "python3 bench.py --steps 1 --warmup 0 --quick" 512x512x90 > /dev/null
python3 tools/pmc_summary.py $O/pmc_sirt $O/pmc_traffic_sirt.txt "python3 tools/run_config.py --alg sirt --iters 1" 512x512x90 > /dev/null
python3 tools/pmc_summary.py $O/pmc_fista $O/pmc_traffic_fista.txt "python3 tools/run_config.py --alg fista --iters 1" 512x512x90 > /dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
tail -1 $O/bench.json | cut -c1-300; head -14 $O/kernel_stats.txt | cut -c1-150; cat $O/fista.log $O/sirt.log | tail -5
