#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 400 python scripts/diag_copies.py > gpurun_out/r6_c29_copies.txt 2>&1
tail -n 60 gpurun_out/r6_c29_copies.txt | cut -c1-200
