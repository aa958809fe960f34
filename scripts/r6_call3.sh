#!/bin/bash
# round 6, call 3: conv tests, gradient split, triangle multiplication sub-batching A/B, kernel stats of the default step
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_triangle_gpu.py -q -x -m gpu > gpurun_out/r6_c3_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 5 gpurun_out/r6_c3_pytest.txt | cut -c1-300
timeout 600 python scripts/diag_grad_split.py network_F32_N256.npz > gpurun_out/r6_grad_split_F32_N256.txt 2>&1 < /dev/null
echo "split rc=$?"; grep -v "Warn\|Consider\|print(\|amdgpu.ids" gpurun_out/r6_grad_split_F32_N256.txt | cut -c1-330
timeout 600 python scripts/diag_grad_split.py network_F8_N512.npz > gpurun_out/r6_grad_split_F8_N512.txt 2>&1 < /dev/null
echo "split rc=$?"; grep -v "Warn\|Consider\|print(\|amdgpu.ids" gpurun_out/r6_grad_split_F8_N512.txt | cut -c1-330
for sub in 0 1 2 4; do
  echo "== DFOLD_TRIMUL_SUB=$sub"
  DFOLD_TRIMUL_SUB=$sub timeout 300 python scripts/bench_triangle.py --n 256 512 --batch 8 --ops tri_mul_out tri_mul_in --no-stages --reps 10 2>&1 | tail -n 6 | cut -c1-300
done
PROF_NAME=r6_kernel_stats bash scripts/gpu_profile.sh > gpurun_out/r6_prof.log 2>&1; tail -n 3 gpurun_out/r6_prof.log | cut -c1-200
