R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/shard64; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --force-dist --quick --nslice 64 --nray 512 --nproj 90 --steps 20 --warmup 2 > $O/log.txt 2>&1
cd $R; python3 tools/prof_summary.py $O/trace $O/stats.txt > /dev/null
tail -1 $O/log.txt | cut -c1-200; head -30 $O/stats.txt | cut -c1-150
python3 tools/gap_analysis.py $O/trace 2>/dev/null | tail -15
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
