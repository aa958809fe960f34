"""Device versions of the two dataset transforms on the hot path's input side:
`atom37_to_frames` and `atom37_to_torsion_angles` of openfold/data/data_transforms.py (:755-893, :923-1088), with the
reference's calling convention (a feature dict in, the same dict updated and returned; the torsion transform is curried
like upstream: `atom37_to_torsion_angles()(protein)`), as used by src/data/Dfold_data_loader_dynamic.py:237-240.
Both run in one HIP launch each (csrc/dataset_geom.hip) on [..., N_res, 37, 3] device tensors; there is no CPU path.
`make_atom14_masks` / `make_atom14_positions` (:572-643, :653-752), the pure index gathers between them in the loader,
are table look-ups on whatever device the features live on."""
from ctypes import c_double, c_int32, c_int64, c_void_p

import torch

from .. import _lib
from ..model.geometry import residue_tables
from ..ops import _p


def _inputs(protein, prefix=""):
    aatype = protein[prefix + "aatype"]
    pos = protein[prefix + "all_atom_positions"]
    mask = protein[prefix + "all_atom_mask"]
    if not pos.is_cuda:
        raise RuntimeError("dynamicpdb_amd.data.data_transforms needs device tensors (no CPU fallback)")
    if pos.shape[-2:] != (37, 3) or mask.shape[-1] != 37 or aatype.shape != pos.shape[:-2]:
        raise ValueError("expected aatype [*, N], all_atom_positions [*, N, 37, 3], all_atom_mask [*, N, 37]")
    N = aatype.shape[-1]
    return (aatype.long().contiguous(), pos.double().contiguous(), mask.expand(pos.shape[:-1]).double().contiguous(), N)


def _launch(aatype, pos, mask, N, frames, torsions, eps=1e-8):
    T = residue_tables(pos.device)
    P = aatype.numel()
    null = c_void_p(0)
    f = [_p(t) for t in frames] if frames else [null] * 5
    t = [_p(x) for x in torsions] if torsions else [null] * 3
    _lib.check(_lib.lib().dfold_atom37_geometry(
        _p(aatype), _p(pos), _p(mask), _p(T["group_base_atom37"]), _p(T["group_mask"]), _p(T["group_ambiguous"]),
        _p(T["chi_atom37"]), _p(T["chi_mask"]), _p(T["chi_pi_periodic"]), *f, *t, c_int64(P), c_int32(N), c_double(eps),
        _lib.stream()), "dfold_atom37_geometry")


def atom37_to_frames(protein, eps=1e-8):
    aatype, pos, mask, N = _inputs(protein)
    lead, dev = tuple(aatype.shape), pos.device
    fr = torch.empty(lead + (8, 4, 4), dtype=torch.float32, device=dev)
    alt = torch.empty_like(fr)
    ex, ge, amb = (torch.empty(lead + (8,), dtype=torch.float64, device=dev) for _ in range(3))
    _launch(aatype, pos, mask, N, (fr, alt, ex, ge, amb), None, eps)
    md = protein["all_atom_mask"].dtype
    protein["rigidgroups_gt_frames"] = fr
    protein["rigidgroups_gt_exists"] = ex.to(md)
    protein["rigidgroups_group_exists"] = ge.to(md)
    protein["rigidgroups_group_is_ambiguous"] = amb.to(md)
    protein["rigidgroups_alt_gt_frames"] = alt
    return protein


def atom37_to_torsion_angles(prefix=""):
    def fn(protein):
        aatype, pos, mask, N = _inputs(protein, prefix)
        lead, dev = tuple(aatype.shape), pos.device
        sc = torch.empty(lead + (7, 2), dtype=torch.float64, device=dev)
        alt = torch.empty_like(sc)
        tm = torch.empty(lead + (7,), dtype=torch.float64, device=dev)
        _launch(aatype, pos, mask, N, None, (sc, alt, tm))
        pd, md = protein[prefix + "all_atom_positions"].dtype, protein[prefix + "all_atom_mask"].dtype
        protein[prefix + "torsion_angles_sin_cos"] = sc.to(pd)
        protein[prefix + "alt_torsion_angles_sin_cos"] = alt.to(pd)
        protein[prefix + "torsion_angles_mask"] = tm.to(md)
        return protein
    return fn


def make_atom14_masks(protein):
    """openfold/data/data_transforms.py:572-643: per-residue atom14 <-> atom37 index maps and existence masks."""
    aatype = protein["aatype"].to(torch.long)
    T = residue_tables(aatype.device)
    protein["atom14_atom_exists"] = T["atom14_exists"][aatype]
    protein["residx_atom14_to_atom37"] = T["atom14_to_atom37"][aatype]
    protein["residx_atom37_to_atom14"] = T["atom37_to_atom14"][aatype]
    protein["atom37_atom_exists"] = T["atom37_mask"][aatype]
    return protein


def make_atom14_positions(protein):
    """openfold/data/data_transforms.py:653-752: atom14 ground-truth positions / masks gathered from atom37, their
    alternatives under the renaming of the chemically equivalent atoms (ASP, GLU, PHE, TYR), and the ambiguity mask.
    The reference multiplies by 14x14 permutation matrices; a permutation is a gather."""
    aatype = protein["aatype"].to(torch.long)
    T = residue_tables(aatype.device)
    exists = protein["atom14_atom_exists"]
    idx = protein["residx_atom14_to_atom37"]
    mask37, pos37 = protein["all_atom_mask"], protein["all_atom_positions"]
    gt_mask = exists * torch.gather(mask37, -1, idx)
    gt_pos = gt_mask[..., None] * torch.gather(pos37, -2, idx[..., None].expand(idx.shape + (3,)))
    ren = T["atom14_rename"][aatype]
    protein["atom14_atom_exists"] = exists
    protein["atom14_gt_exists"] = gt_mask
    protein["atom14_gt_positions"] = gt_pos
    protein["atom14_alt_gt_positions"] = torch.gather(gt_pos, -2, ren[..., None].expand(ren.shape + (3,)))
    protein["atom14_alt_gt_exists"] = torch.gather(gt_mask, -1, ren)
    protein["atom14_atom_is_ambiguous"] = T["atom14_is_ambiguous"][aatype].to(mask37.dtype)
    return protein
