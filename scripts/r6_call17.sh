#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 300 python scripts/triatt_phase_times.py 256 > gpurun_out/r6_c17_phases.txt 2>&1
cat gpurun_out/r6_c17_phases.txt | cut -c1-1200
