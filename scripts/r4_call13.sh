set -u
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_ipa_gpu.py tests/test_network_gpu.py "tests/test_parity_baseline_gpu.py::test_step_vs_reference_golden_config1" "tests/test_parity_baseline_gpu.py::test_res_mask_holes_vs_reference_golden" -x -q -m gpu 2>&1 | tail -n 4 | cut -c1-300 )
for v in 1 0 1 0; do
DFOLD_PAIR_PROJ_FUSED=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-triangle --no-last-frame-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ms,c,w=d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['second_kernel']['avg_launch_ms']
print('pairproj=$v', ms, c, w, 'non-conv', round(ms-64*c-32*w,2))"
done
