"""GPU parity of the OmegaFold pair-track drop-ins (dynamicpdb_amd/model/geoformer.py: Node2Edge, GeometricAttention)
against golden vectors minted from the reference's vendored OmegaFold modules (tests/golden/geoformer_S5_N24.npz,
src/toolbox/OmegaFold/omegafold/modules.py:320-351,568-723) and, at N_res = 80 (not a multiple of 64: ragged tiles of the
fused kernels) with a batch axis, against the CPU oracle.  bf16 operands / fp32 accumulation: 1.5e-2 relative L2."""
import numpy as np
import pytest
import torch

from util import load_golden, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(mod, g, prefix):
    mod.load_state_dict({k[len(prefix):]: torch.tensor(v) for k, v in g.items() if k.startswith(prefix)}, strict=True)
    return mod.to(DEV)


def test_node2edge_vs_reference_golden():
    from dynamicpdb_amd.model.geoformer import Node2Edge
    g = load_golden("geoformer_S5_N24.npz")
    m = _load(Node2Edge(in_dim=256, proj_dim=32, out_dim=128), g, "n2e.P.")
    with torch.no_grad():
        y = m(torch.tensor(g["node"]).to(DEV), torch.tensor(g["seq_mask"]).to(DEV))
    assert y.shape == g["n2e.out"].shape and rel_l2(y, g["n2e.out"]) < 1.5e-2, rel_l2(y, g["n2e.out"])
    with pytest.raises(RuntimeError):          # inference-only: refuses to run where a gradient would be expected
        m(torch.tensor(g["node"]).to(DEV), torch.tensor(g["seq_mask"]).to(DEV))


def test_geometric_attention_vs_reference_golden():
    from dynamicpdb_amd.model.geoformer import GeometricAttention
    g = load_golden("geoformer_S5_N24.npz")
    m = _load(GeometricAttention(d_edge=128, c=32, n_head=4, n_axis=2), g, "ga.P.")
    assert sorted(k for k, _ in m.named_parameters()) == sorted(k[5:] for k in g if k.startswith("ga.P."))
    with torch.no_grad():
        y = m(torch.tensor(g["edge"]).to(DEV), torch.tensor(g["res_mask"]).to(DEV), None)
    err = rel_l2(y, g["ga.out"])
    assert y.shape == g["ga.out"].shape and err < 1.5e-2, err
    # transpose-detecting: the result is not symmetric and a swapped axis pair would be far off
    assert rel_l2(y.transpose(0, 1), g["ga.out"]) > 0.3


def test_geometric_attention_batched_ragged_vs_oracle():
    from oracle import dfold_oracle as O
    from dynamicpdb_amd.model.geoformer import GeometricAttention
    rng = np.random.default_rng(31)
    m = GeometricAttention(d_edge=128, c=32, n_head=4, n_axis=2)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.tensor((rng.standard_normal(tuple(p.shape)) * 0.09).astype(np.float32)))
    P = {k: v.detach().clone() for k, v in m.named_parameters()}
    m.to(DEV)
    B, N = 2, 80
    e = torch.tensor((rng.standard_normal((B, N, N, 128)) * 1.3).astype(np.float32))
    mask = torch.tensor((rng.uniform(size=(B, N)) > 0.1).astype(np.float32))
    with torch.no_grad():
        y = m(e.to(DEV), mask.to(DEV), None).cpu()
        for b in range(B):
            ref = O.omegafold_geometric_attention(P, e[b], mask[b])
            assert rel_l2(y[b], ref) < 1.5e-2, (b, rel_l2(y[b], ref))
