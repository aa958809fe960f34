# round 5, call 5: plain-torch training step of the drop-in network (no engine trainer) + dump for the reference-side loss_fn check
set -u
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_training_gpu.py -q -x -k "plain_torch_step" 2>&1 | tail -n 6 ) | cut -c1-300
ls -la gpurun_out/dropin_step_F3_N16.npz
