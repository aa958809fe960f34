# round-4 profile passes on one GPU box (every pass bounded): kernel-trace stats of the step, PMC passes for the conv kernels,
# the IPA attention kernels and the triangle operators (forward + streaming backward)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PROF_NAME=r4_kernel_stats bash scripts/gpu_profile.sh > gpurun_out/r4_prof_step.log 2>&1; tail -n 2 gpurun_out/r4_prof_step.log | cut -c1-200
PROF_TAG=r4 bash scripts/gpu_pmc.sh > gpurun_out/r4_prof_pmc.log 2>&1; tail -n 4 gpurun_out/r4_prof_pmc.log | cut -c1-300
PROF_TAG=r4 bash scripts/gpu_ipa_pmc.sh > gpurun_out/r4_prof_ipa.log 2>&1; grep -c ipa_ gpurun_out/r4_prof_ipa.log
PROF_TAG=r4 TRI_ARGS="--n 256 512 --reps 5 --backward" PMC_ARGS="--n 256 --reps 2 --no-stages --backward --batch 8" PASSES="fetch write" bash scripts/gpu_triangle_profile.sh > gpurun_out/r4_prof_tri.log 2>&1; tail -n 3 gpurun_out/r4_prof_tri.log | cut -c1-300
timeout 300 python scripts/bench_triangle.py --n 256 512 --batch 1 --backward > gpurun_out/r4_triangle_bench_b1.jsonl 2>/dev/null
timeout 300 python scripts/bench_triangle.py --n 256 512 --batch 8 --backward > gpurun_out/r4_triangle_bench_b8.jsonl 2>/dev/null
wc -l gpurun_out/r4_triangle_bench_b*.jsonl
