# What-if for tv_gd temporal blocking: the update pass of k_tv_march4 ALSO does the norm pass's arithmetic of the next
# iteration on the rows of x_new it holds in registers (results wrong: no ring columns, no edge marches) -- the LOWER bound of a fused kernel.
# Build first:  make -C tomo_tv_amd/csrc OUT=../libtomo_whatif.so EXTRA="-DTV4_WHATIF_FUSED -DTV4_UPD_WAVES=3"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tvwf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prod -- python3 $R/tools/bench_tv.py --reps 5 > $O/prod.log 2>&1
TOMO_LIB=$R/tomo_tv_amd/libtomo_whatif.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/wf -- python3 $R/tools/bench_tv.py --reps 5 > $O/wf.log 2>&1
cd $R
python3 tools/prof_summary.py $O/prod $O/prod_stats.txt > /dev/null; python3 tools/prof_summary.py $O/wf $O/wf_stats.txt > /dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/prod.log $O/wf.log | grep tv_gd; grep k_tv_march4 $O/prod_stats.txt | cut -c1-200; echo; grep k_tv_march4 $O/wf_stats.txt | cut -c1-200
# in the ASD-POCS step: production, what-if with its norm passes still launched, what-if with them skipped
RUN='import os,sys,runpy; from tomo_tv_amd import _lib; _lib.LIB_PATH=os.environ.get("TOMO_LIB", _lib.LIB_PATH); sys.argv=["bench.py","--quick","--no-cpu-baseline"]; runpy.run_path("bench.py", run_name="__main__")'
for i in 1 2; do
python3 -c "$RUN" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('production   ms_per_step', d['ms_per_step'])"
TOMO_LIB=$R/tomo_tv_amd/libtomo_whatif.so python3 -c "$RUN" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('what-if + norm passes ms_per_step', d['ms_per_step'])"
TOMO_WHATIF_SKIP_NORM=1 TOMO_LIB=$R/tomo_tv_amd/libtomo_whatif.so python3 -c "$RUN" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('what-if, norm passes skipped ms_per_step', d['ms_per_step'])"
done
