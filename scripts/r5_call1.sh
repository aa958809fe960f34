# round 5, call 1: the one-wave-per-SIMD conv kernel (csrc/conv_fwd_w4.hip): parity, then A/B timing against the 256x320 halo kernel
set -u
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gemm_gpu.py -q -x -k "one_wave_per_simd or conv_fwd_large or conv_tower_last_frame_cone" 2>&1 | tail -n 15 ) | cut -c1-300
for v in 1 0 1 0; do
  echo "DFOLD_CONV_W4=$v"
  DFOLD_CONV_W4=$v timeout 300 python scripts/exp_conv_dvfs.py 2>&1 | tail -n 2 | cut -c1-300
done
for v in 1 0; do
  echo "epilogue probe DFOLD_CONV_W4=$v"
  DFOLD_CONV_W4=$v timeout 300 python scripts/bench_conv.py epilogue 2>&1 | tail -n 5 | cut -c1-300
done
