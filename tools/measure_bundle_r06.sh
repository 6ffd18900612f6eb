# Measurement bundle of round 6 (run on the GPU box through gpurun: bash tools/measure_bundle_r06.sh): the driver's bench command, rocprofv3 kernel
# stats of the same command and of the FISTA / SIRT / sharded runs, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs, no tracing domains
# besides the kernel trace).  Summaries land in gpurun_out/r06/; copy what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT; TAG=r06; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err   # the driver's command
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --quick --steps 20 --warmup 5 > $O/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fista -- python3 $R/tools/run_config.py --alg fista --iters 5 > $O/fista.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sirt -- python3 $R/tools/run_config.py --alg sirt --iters 10 > $O/sirt.log 2>&1
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_shard64 -- python3 $R/bench.py --force-dist --quick --nslice 64 --nray 512 --nproj 90 --steps 20 --warmup 2 > $O/shard64.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$grp -- python3 $R/bench.py --steps 1 --warmup 0 --quick > $O/pmc_$grp.log 2>&1; done
for grp in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $grp --output-format csv -d $O/pmcs_$grp -- python3 $R/tools/run_config.py --alg sirt --iters 1 > $O/pmcs_$grp.log 2>&1; done
cd $R
python3 tools/prof_summary.py $O/trace $O/kernel_stats.txt > /dev/null
python3 tools/prof_summary.py $O/trace_fista $O/kernel_stats_fista.txt > /dev/null
python3 tools/prof_summary.py $O/trace_sirt $O/kernel_stats_sirt.txt > /dev/null
python3 tools/prof_summary.py $O/trace_shard64 $O/kernel_stats_shard64.txt > /dev/null
mkdir -p $O/pmc_bench $O/pmc_sirt; mv $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_bench/; mv $O/pmcs_FETCH_SIZE $O/pmcs_WRITE_SIZE $O/pmc_sirt/
python3 tools/pmc_summary.py $O/pmc_bench $O/pmc_traffic.txt "python3 bench.py --steps 1 --warmup 0 --quick" 512x512x90 > /dev/null
python3 tools/pmc_summary.py $O/pmc_sirt $O/pmc_traffic_sirt.txt "python3 tools/run_config.py --alg sirt --iters 1" 512x512x90 > /dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
tail -1 $O/bench.json | cut -c1-300; head -14 $O/kernel_stats.txt | cut -c1-150; cat $O/fista.log $O/sirt.log | tail -5; tail -1 $O/shard64.log | cut -c1-200
