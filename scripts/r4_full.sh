set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^\s*$" | tail -n 150 ) > gpurun_out/r4_pytest_gpu_full.log 2>&1
tail -n 12 gpurun_out/r4_pytest_gpu_full.log | cut -c1-300
