"""Micro-benchmark of the IPA attention core forward at BASELINE config 3 shapes (B=8, F=32, N=256, H=8, C=256):
fused kernel (csrc/ipa_fused.hip) with / without the fp32 copy of the probabilities, and the unfused round-2 chain.
Run on the GPU box:  python scripts/bench_ipa.py [N]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dynamicpdb_amd.model import functional as Fm  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    B, F, H, C, CZ, PZ = (8, 32, 8, 256, 128, 32) if N <= 256 else (2, 64, 8, 256, 128, 32)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    q = rn(B, F, N, H * C).to(torch.bfloat16)
    kv = rn(B, F, N, 2 * H * C).to(torch.bfloat16)
    steps = torch.randn(B, F, N, 3, generator=g)
    chain = torch.cumsum(3.8 * steps / steps.norm(dim=-1, keepdim=True), 2).to(dev)[:, :, :, None, None, :]
    q_pts, k_pts, v_pts = chain + rn(B, F, N, H, 8, 3), chain + rn(B, F, N, H, 8, 3), chain + rn(B, F, N, H, 12, 3)
    z = rn(B, N, N, CZ).to(torch.bfloat16)
    w_b, w_dz, b_dz = rn(H, CZ, scale=0.1), rn(PZ, CZ, scale=0.1), rn(PZ, scale=0.1)
    mask = torch.ones(B, F, N, device=dev)
    hw = torch.full((H,), 0.06, device=dev)
    args = (q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz, mask, hw)

    def run(tag, fused, keep32, reps=5):
        Fm._IPA_FUSED, Fm._IPA_KEEP_P32 = fused, keep32
        with torch.no_grad():
            for _ in range(2):
                Fm.IpaCoreFn.apply(*args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                Fm.IpaCoreFn.apply(*args)
            e1.record()
            torch.cuda.synchronize()
        print(f"{tag}: {e0.elapsed_time(e1) / reps:.3f} ms per IpaCoreFn forward (incl. pair projections + o_pair)", flush=True)

    if "--fwdbwd" in sys.argv:       # one forward + backward of the node (the PMC passes of scripts/gpu_ipa_pmc.sh profile this)
        Fm._IPA_FUSED, Fm._IPA_KEEP_P32 = True, False
        leaves = [t.clone().requires_grad_(True) for t in (q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz)] + [mask, hw.clone().requires_grad_(True)]
        for rep in range(2):
            o, o_pt, o_pair = Fm.IpaCoreFn.apply(*leaves)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.autograd.backward([o, o_pt, o_pair], [torch.ones_like(o), torch.ones_like(o_pt), torch.ones_like(o_pair)])
            e1.record()
            torch.cuda.synchronize()
            print(f"IpaCoreFn backward: {e0.elapsed_time(e1):.3f} ms", flush=True)
        return
    run("unfused chain", False, True)
    run("fused, fp32 P kept", True, True)
    run("fused, bf16 P only", True, False)


if __name__ == "__main__":
    main()
