#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
for v in 1 0 1; do
  DFOLD_IPA_PAIR_STREAM=$v timeout 300 python -m pytest tests/test_training_gpu.py -q -x -m gpu -k "two_ranks_on_one_gpu" > gpurun_out/r6_c8_pytest_$v.txt 2>&1 < /dev/null
  echo "pair_stream=$v rc=$?"; tail -n 3 gpurun_out/r6_c8_pytest_$v.txt | cut -c1-200
done
