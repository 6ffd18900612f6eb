#!/bin/bash
# The phase timeline of the volume-resident SART sweep (k_sart_resident): builds tools/experiments/resident_probe.hip twice -- plain
# (the timing the bench's roofline record must reproduce) and with -DRS_PROF (s_memrealtime stamps per phase, wave 0 of every
# workgroup, and the per-wave timeline of one step of one workgroup) -- and runs both on the headline geometry (512^2 x 90,
# one 64-slice chunk = every CU busy) and on the whole 512-slice slab.  Run on the GPU box:
#   gpurun -- 'bash tools/resident_phases.sh > gpurun_out/r06_resident_phases.txt 2>&1'
# and copy the output to profiles/ (bench.py's roofline.frac for k_sart_resident follows from the "us per angle and chunk" here).
set -e
cd "$(dirname "$0")/experiments"
CS=../../tomo_tv_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -I$CS"
/opt/rocm/bin/hipcc $FLAGS resident_probe.hip $CS/sysmat.cpp $CS/resident.cpp -lpthread -o resident_probe_np
/opt/rocm/bin/hipcc $FLAGS -DRS_PROF resident_probe.hip $CS/sysmat.cpp $CS/resident.cpp -lpthread -o resident_probe
echo "== plain build: 512^2 x 90, 64 slices (one chunk), untracked / tracked; 512 slices tracked (the headline's sweep)"
./resident_probe_np 512 90 64 1 5 | tail -2
./resident_probe_np 512 90 64 1 5 1 | tail -3
./resident_probe_np 512 90 512 1 3 1 | tail -3
echo "== profiling build (RS_PROF): phase totals and the timeline of step 40, workgroup 37"
./resident_probe 512 90 64 1 5 | grep -v '^rep\|^slice'
echo "== bytes: per angle and 64-slice chunk the kernel must move 16 B x 512^2 of cells; per chunk 2 x 64 x 512^2 x 4 B of slab (+ 2 x the same, tracked) and 90 x 512 x 64 x 4 B of measured rows"
