#!/bin/bash
# Builds bayesianfiltering_amd/libbayesfilt_timers.so: the regular library with BF_MFMA_PHASE_TIMERS compiled into the
# MFMA Kalman kernel (per-phase wall-clock ticks, read back by scripts/mfma_phase_probe.py).
set -e
cd "$(dirname "$0")/../bayesianfiltering_amd/csrc"

/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc -DBF_MFMA_PHASE_TIMERS $BF_EXTRA -c kf_scan_mfma.hip -o /tmp/kf_scan_mfma_timers.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libbayesfilt_timers.so $(ls *.o | grep -v '^kf_scan_mfma.o$') /tmp/kf_scan_mfma_timers.o
