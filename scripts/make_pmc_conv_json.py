"""profiles/<tag>_pmc_conv.json (what bench.py reads for roofline.traffic) from the two machine-readable PMC summaries
written by scripts/gpu_pmc.sh:  python scripts/make_pmc_conv_json.py r4"""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out") if os.path.exists(os.path.join(root, "gpurun_out", f"{tag}_pmc_fetch_hit.json")) else os.path.join(root, "profiles")
a = json.load(open(os.path.join(src, f"{tag}_pmc_fetch_hit.json")))
b = json.load(open(os.path.join(src, f"{tag}_pmc_write_miss_req.json")))


def pick(d, key):
    k = next(k for k in d if key in k)
    return d[k]


def entry(key, name):
    f, w = pick(a, key), pick(b, key)
    return {"kernel": name, "launches_averaged": int(f["launches"]), "FETCH_SIZE_kb": f["FETCH_SIZE"], "WRITE_SIZE_kb": w["WRITE_SIZE"],
            "TCC_REQ": w["TCC_REQ_sum"], "TCC_HIT": f["TCC_HIT_sum"], "TCC_MISS": w["TCC_MISS_sum"]}


try:
    try:        # round 6: the plain instantiation = the forward launches (the NZ instantiation skips tiles)
        out = entry("dfold_conv_w4_kernel<false, false>", "dfold_conv_w4_kernel<false, false> (one wave per SIMD, 512 x 160 tile; forward launches)")
        out["nz_instantiation"] = entry("dfold_conv_w4_kernel<false, true>", "dfold_conv_w4_kernel<false, true> (data-gradient launches with zero-frame flags)")
    except StopIteration:
        out = entry("dfold_conv_w4_kernel", "dfold_conv_w4_kernel (one wave per SIMD, 512 x 160 tile)")
except StopIteration:      # passes taken with DFOLD_CONV_W4=0 / before round 5
    out = entry("dfold_mfma_gemm320_kernel<1, 5, true>", "dfold_mfma_gemm320_kernel<1, 5, true> (halo form)")
out["note"] = ("rocprofv3 --pmc, two passes (FETCH_SIZE TCC_HIT_sum | WRITE_SIZE TCC_MISS_sum TCC_REQ_sum; scripts/gpu_pmc.sh -> "
               f"profiles/{tag}_pmc_fetch_hit.txt, {tag}_pmc_write_miss_req.txt), averages per launch over the conv forward+dgrad launches of "
               "the all-frames update_fn steps of the pass at BASELINE config 3; FETCH_SIZE is doubled when converted to bytes (gfx950 "
               "wide-coalesced-read correction, MI355X_MICROARCH.md section HBM); WRITE_SIZE is uncalibrated")
try:
    out["wgrad_tn"] = entry("conv_wgrad_tn_kernel<false>", "conv_wgrad_tn_kernel<false>")
except StopIteration:
    out["wgrad_tn"] = entry("conv_wgrad_tn_kernel", "conv_wgrad_tn_kernel")
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc_conv.json"), "w"), indent=1)
print(json.dumps(out)[:600])
