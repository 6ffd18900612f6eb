#!/bin/bash
# What-if (results WRONG, measurement only; nothing of this lives in the product tree): is the HBM traffic of the all-angle projector pair
# the lever on its time?  VERDICT r5 item 3: k_fp_list moves 7.3 x, k_bp_list 4.3 x their algorithmic bytes (the volume staged once per
# pass of ~16 angles; the residual windows of 16 x 16 tiles) -- "a what-if build that shows the pass count is not the lever also closes it".
# Builds, from a patched COPY of tomo_tv_amd/csrc:
#   A  k_fp_list stages every tile from the same 4096 pixels (8 MB: they stay in the L2 / Infinity Cache)  -> no volume traffic at all
#   B  k_bp_list stages every residual window from the first 64 rows of the sinogram                       -> no window traffic at all
#   C  k_bp_list without its epilogue's read of x (alpha = 0 path forced) -- the 0.54 GB the epilogue reads behind the last stage
# and times FP / SIRT at 512^3 x 90 with each (tools/bench_fp.py, TOMO_LIB = the patched build) next to the product build, three
# rounds interleaved.   gpurun -- 'bash tools/experiments/projector_traffic_whatif.sh > gpurun_out/r06_projector_whatif.txt 2>&1'
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
W=$R/gpurun_out/projwf; rm -rf $W; mkdir -p $W
build() {   # build <name> <sed expression on kernels_fp.hip.h> <sed expression on kernels_bp.hip.h>
  rm -rf $W/src_$1; mkdir -p $W/src_$1/tomo_tv_amd $W/src_$1/include; cp -r $R/tomo_tv_amd/csrc $W/src_$1/tomo_tv_amd/; cp $R/include/*.h $W/src_$1/include/
  [ -n "$2" ] && sed -i "$2" $W/src_$1/tomo_tv_amd/csrc/kernels_fp.hip.h
  [ -n "$3" ] && sed -i "$3" $W/src_$1/tomo_tv_amd/csrc/kernels_bp.hip.h
  make -s -C $W/src_$1/tomo_tv_amd/csrc OUT=$W/lib_$1.so 2>&1 | grep -v warning | grep -i error || true
  ls -la $W/lib_$1.so | awk '{print "built", $NF, $5, "bytes"}'
}
build A 's|const float \*src = ok ? xc + pix \* sx : zsrc;|const float *src = ok ? xc + (pix \& 4095) * sx : zsrc;|' ''
grep -c 'pix & 4095' $W/src_A/tomo_tv_amd/csrc/kernels_fp.hip.h
build B '' 's|(rc + ((size_t)((S) \* BL_A + a) \* n + (ww \& 0xFFFFu) + 2 \* pr) \* sx)|(rc + ((size_t)((2 * pr) \& 63)) * sx)|'
grep -c '(2 \* pr) & 63' $W/src_B/tomo_tv_amd/csrc/kernels_bp.hip.h
build C '' 's|if (alpha != 0.f) {|if (false) {|; s|if (alpha != 0.f) nv = bp_axpby(alpha, xv\[k\], beta, a);||'
cd $R
for i in 1 2 3; do
  echo "== round $i"
  echo -n "product            "; python3 tools/bench_fp.py --reps 20 | tail -1
  for v in A B C; do echo -n "what-if $v          "; TOMO_LIB=$W/lib_$v.so python3 tools/bench_fp.py --reps 20 | tail -1; done
done
