#!/bin/bash
# Compile the MFMA Kalman kernel to assembly and print register / spill usage and the loop instruction mix.
cd /root/repo/bayesianfiltering_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I. -I../../include -S --cuda-device-only kf_scan_mfma.hip -o /tmp/m.s -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "error|VGPRs|Scratch|Spill"
python /root/repo/scripts/asm_loops.py /tmp/m.s | head -1
