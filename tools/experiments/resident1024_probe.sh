#!/bin/bash
# Compute-side probe of a register-resident SART sweep at N = 1024 (tools/experiments/resident1024_probe.hip).
#   gpurun -- 'bash tools/experiments/resident1024_probe.sh > gpurun_out/r06_resident1024_probe.txt 2>&1'
set -e
cd "$(dirname "$0")"
CS=../../tomo_tv_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$CS resident1024_probe.hip $CS/sysmat.cpp -lpthread -o resident1024_probe
echo "== small geometry first (512^2 x 24: 64 tiles), full step, bit-compared with the CPU replay"
./resident1024_probe 512 24 24 2 3 | tail -4
echo "== 1024^2 x 120, one sweep: full step (mode 3, bit-compared), back projection only (1), forward projection only (2), forward without its LDS adds (4)"
for m in 3 1 2 4; do ./resident1024_probe 1024 120 120 3 $m | grep -v "^rep"; done
echo "== the forward projection WITHOUT atomics (run totals by three DPP shifts, plain read - add - write by the run's last lane): full step (mode 9, bit-compared), forward only (8)"
./resident1024_probe 512 24 24 2 9 | tail -3
for m in 9 8; do ./resident1024_probe 1024 120 120 3 $m | grep -v "^rep"; done
