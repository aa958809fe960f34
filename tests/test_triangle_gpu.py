"""GPU parity of the triangle pair operators against golden vectors minted from the reference's vendored OpenFold
modules (tests/golden/triangle_N24.npz: outputs, input gradients and every parameter gradient), and against the CPU
oracle at a second size.  bf16 operands / fp32 accumulation: tolerances as in DESIGN.md."""
import numpy as np
import pytest
import torch

from util import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _mods():
    from dynamicpdb_amd.model import triangle as T
    return dict(tri_mul_out=lambda: T.TriangleMultiplicationOutgoing(128, 128),
                tri_mul_in=lambda: T.TriangleMultiplicationIncoming(128, 128),
                tri_att_start=lambda: T.TriangleAttentionStartingNode(128, 32, 4),
                tri_att_end=lambda: T.TriangleAttentionEndingNode(128, 32, 4))


@pytest.mark.parametrize("name", ["tri_mul_out", "tri_mul_in", "tri_att_start", "tri_att_end"])
def test_triangle_vs_reference_golden(name):
    dev = torch.device("cuda:0")
    g = load_golden("triangle_N24.npz")
    m = _mods()[name]()
    sd = {k[len(name) + 3:]: torch.tensor(v) for k, v in g.items() if k.startswith(name + ".P.")}
    m.load_state_dict(sd, strict=True)          # same state_dict keys / shapes as the OpenFold module
    m.to(dev)
    z = torch.tensor(g["z"]).to(dev).requires_grad_(True)
    mask = torch.tensor(g["mask"]).to(dev)
    y = m(z, mask=mask)
    assert rel_l2(y, g[name + ".out"]) < 1.5e-2
    y.backward(torch.tensor(g[name + ".gy"]).to(dev))
    assert rel_l2(z.grad, g[name + ".gz"]) < 3e-2
    for k, p in m.named_parameters():
        ref = g[f"{name}.G.{k}"]
        assert p.grad is not None, k
        assert rel_l2(p.grad, ref) < 3e-2, (k, rel_l2(p.grad, ref))


@pytest.mark.parametrize("name", ["tri_mul_out", "tri_mul_in", "tri_att_start", "tri_att_end"])
def test_triangle_vs_oracle_n64(name):
    from oracle import dfold_oracle as O
    dev = torch.device("cuda:0")
    N = 64
    rng = np.random.default_rng(11)
    m = _mods()[name]()
    sd = m.state_dict()
    for k, v in sd.items():
        if v.dim() >= 2:
            w = rng.standard_normal(tuple(v.shape), dtype=np.float32) / np.sqrt(v.shape[-1])
        elif k.endswith("weight"):
            w = 1.0 + 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
        else:
            w = 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
        sd[k] = torch.tensor(w)
    m.load_state_dict(sd)
    m.to(dev)
    z = torch.tensor(rng.standard_normal((N, N, 128), dtype=np.float32))
    mask = torch.tensor((rng.uniform(size=(N, N)) > 0.1).astype(np.float32))
    with torch.no_grad():
        y = m(z.to(dev), mask=mask.to(dev))
    P = {k: v for k, v in sd.items()}
    if name.startswith("tri_mul"):
        ref = O.triangle_multiplication(P, z, mask, outgoing=name.endswith("out"))
    else:
        ref = O.triangle_attention(P, z, mask, starting=name.endswith("start"))
    assert rel_l2(y, ref) < 1.5e-2


def test_pair_transition_vs_reference_golden_and_oracle():
    """PairTransition against the reference module's golden (N = 24) and against the oracle at N = 96 with a batch axis."""
    from oracle import dfold_oracle as O
    from dynamicpdb_amd.model import triangle as T
    dev = torch.device("cuda:0")
    g = load_golden("pair_transition_N24.npz")
    m = T.PairTransition(128, 4)
    m.load_state_dict({k[2:]: torch.tensor(v) for k, v in g.items() if k.startswith("P.")}, strict=True)
    m.to(dev)
    z = torch.tensor(g["z"]).to(dev).requires_grad_(True)
    y = m(z, mask=torch.tensor(g["mask"]).to(dev))
    assert rel_l2(y, g["out"]) < 1e-2
    y.backward(torch.tensor(g["gy"]).to(dev))
    # gradient path: three bf16 storage points (dL/dy, dL/dh, dL/dx) and a ReLU mask on a bf16-rounded activation
    assert rel_l2(z.grad, g["gz"]) < 5e-2
    for k, p in m.named_parameters():
        assert p.grad is not None and rel_l2(p.grad, g["G." + k]) < 5e-2, (k, rel_l2(p.grad, g["G." + k]))
    rng = np.random.default_rng(2)
    z2 = torch.tensor(rng.standard_normal((2, 96, 96, 128), dtype=np.float32))
    mask2 = torch.tensor((rng.uniform(size=(2, 96, 96)) > 0.2).astype(np.float32))
    with torch.no_grad():
        y2 = m(z2.to(dev), mask=mask2.to(dev), chunk_size=4)
        ref = O.pair_transition({k: v.detach().cpu() for k, v in m.state_dict().items()}, z2, mask2)
    assert rel_l2(y2, ref) < 1e-2
    assert float(y2.cpu()[mask2 == 0].abs().max()) == 0.0
