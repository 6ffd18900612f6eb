# A/B of two builds of the library on the TV descent alone: per-kernel durations (tools/ab_tv_profile.sh <other .so> [bench_tv args])
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ab_tv; OTHER=$1; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/new -- python3 $R/tools/bench_tv.py --reps 3 "$@" > $O/new.log 2>&1
export TOMO_LIB=$R/$OTHER
rocprofv3 --kernel-trace --stats --output-format csv -d $O/old -- python3 $R/tools/bench_tv.py --reps 3 "$@" > $O/old.log 2>&1
cd $R
for w in new old; do echo "== $w"; cat $O/$w.log | tail -1; f=$(find $O/$w -name "*kernel_stats.csv" | head -1); head -7 $f | cut -d, -f1-4,6-8 | cut -c1-160; done
