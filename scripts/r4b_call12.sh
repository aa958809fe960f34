# round 4 (second session), call 12: counters of the query-block triangle-attention kernel at batch 8 x N_res 512 (SQ issue / stall
# buckets, HBM-side bytes) -- separate --pmc passes
set -u
cd $GRAFT_REPO_ROOT
PROF_TAG=r4b TRI_ARGS="--ops tri_att_start --n 512 --batch 8 --reps 3 --no-stages" PMC_ARGS="--ops tri_att_start --n 512 --batch 8 --reps 2 --no-stages" bash scripts/gpu_triangle_profile.sh 2>&1 | grep -v "^E2026\|^W2026" | cut -c1-300 | tail -n 30
for t in sq sq2 fetch write; do echo "== $t"; grep -i "triatt_rows\|tri_bias\|Name\|kernel" gpurun_out/r4b_triangle_pmc_$t.txt | head -n 6 | cut -c1-600; done
