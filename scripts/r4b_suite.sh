# full GPU suite on the final code of the round, with the printed diagnostics
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^\s*$" | tail -n 200 ) > gpurun_out/r4_pytest_gpu.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r4_pytest_gpu.txt | tail -n 3 | cut -c1-200
