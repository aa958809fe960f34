"""TEST/BUILD INFRASTRUCTURE: dump the numeric residue-geometry tables the
frames->atom14/atom37 step indexes by aatype (reference: src/data/all_atom.py:13-23,
src/model/Dfold_network_dynamic.py:574-594 via openfold/np/residue_constants.py).
Only numbers are exported (ideal-geometry constants), no code.

Run here (needs /root/reference):  python oracle/ref_harness/dump_residue_tables.py
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import
ref_import.install()
from src.data import residue_constants as rc_src
from openfold.np import residue_constants as rc_of

# --- index / mask tables of the dataset-side geometry (openfold/data/data_transforms.py:755-893 atom37_to_frames,
#     :895-1088 atom37_to_torsion_angles), built from the same residue constants the reference loops over there ---
base_names = np.full([21, 8, 3], "", dtype=object)
base_names[:, 0, :] = ["C", "CA", "N"]
base_names[:, 3, :] = ["CA", "C", "O"]
group_mask = np.zeros([21, 8], np.float32)
group_mask[:, 0] = 1
group_mask[:, 3] = 1
group_mask[:20, 4:] = np.asarray(rc_of.chi_angles_mask, np.float32)
for restype, letter in enumerate(rc_of.restypes):
    resname = rc_of.restype_1to3[letter]
    for chi in range(4):
        if rc_of.chi_angles_mask[restype][chi]:
            base_names[restype, chi + 4, :] = rc_of.chi_angles_atoms[resname][chi][1:]
lut = dict(rc_of.atom_order)
lut[""] = 0
group_base_atom37 = np.vectorize(lambda x: lut[x])(base_names).astype(np.int64)                  # [21,8,3]
group_ambiguous = np.zeros([21, 8], np.float32)
for resname in rc_of.residue_atom_renaming_swaps:
    restype = rc_of.restype_order[rc_of.restype_3to1[resname]]
    group_ambiguous[restype, int(sum(rc_of.chi_angles_mask[restype]) - 1) + 4] = 1
chi_atoms = []
for letter in rc_of.restypes:
    rows = [[rc_of.atom_order[a] for a in chi] for chi in rc_of.chi_angles_atoms[rc_of.restype_1to3[letter]]]
    rows += [[0, 0, 0, 0]] * (4 - len(rows))
    chi_atoms.append(rows)
chi_atoms.append([[0, 0, 0, 0]] * 4)
chi_mask = np.asarray(list(rc_of.chi_angles_mask) + [[0.0, 0.0, 0.0, 0.0]], np.float32)            # [21,4]
chi_pi = np.asarray(rc_of.chi_pi_periodic, np.float32)                                            # [21,4]

# --- atom14 <-> atom37 maps and the ambiguous-atom renaming of make_atom14_masks / make_atom14_positions
#     (openfold/data/data_transforms.py:572-643, :653-752) ---
a14_to_a37, a14_exists = [], []
for letter in rc_of.restypes:
    names = rc_of.restype_name_to_atom14_names[rc_of.restype_1to3[letter]]
    a14_to_a37.append([(rc_of.atom_order[n] if n else 0) for n in names])
    a14_exists.append([(1.0 if n else 0.0) for n in names])
a14_to_a37.append([0] * 14)
a14_exists.append([0.0] * 14)
a14_rename = np.tile(np.arange(14, dtype=np.int64), (21, 1))
a14_ambiguous = np.zeros([21, 14], np.float32)
for resname, swap in rc_of.residue_atom_renaming_swaps.items():
    restype = rc_of.restype_order[rc_of.restype_3to1[resname]]
    names = rc_of.restype_name_to_atom14_names[resname]
    for n1, n2 in swap.items():
        i1, i2 = names.index(n1), names.index(n2)
        a14_rename[restype, i1], a14_rename[restype, i2] = i2, i1
        a14_ambiguous[restype, i1] = a14_ambiguous[restype, i2] = 1

out = os.path.join(os.path.dirname(__file__), "..", "..", "dynamicpdb_amd", "data", "residue_tables.npz")
np.savez_compressed(
    out,
    group_base_atom37=group_base_atom37, group_mask=group_mask, group_ambiguous=group_ambiguous,
    chi_atom37=np.asarray(chi_atoms, np.int64), chi_mask=chi_mask, chi_pi_periodic=chi_pi,
    atom14_to_atom37=np.asarray(a14_to_a37, np.int64), atom14_exists=np.asarray(a14_exists, np.float32),
    atom14_rename=a14_rename, atom14_is_ambiguous=a14_ambiguous,
    default_frames=np.asarray(rc_src.restype_rigid_group_default_frame, np.float32),      # [21,8,4,4]
    atom14_group=np.asarray(rc_src.restype_atom14_to_rigid_group, np.int64),             # [21,14]
    atom14_mask=np.asarray(rc_src.restype_atom14_mask, np.float32),                      # [21,14]
    atom14_pos=np.asarray(rc_src.restype_atom14_rigid_group_positions, np.float32),      # [21,14,3]
    atom37_to_atom14=np.asarray(rc_of.RESTYPE_ATOM37_TO_ATOM14, np.int64),               # [21,37]
    atom37_mask=np.asarray(rc_of.RESTYPE_ATOM37_MASK, np.float32),                       # [21,37]
)
d = np.load(out)
for k in d.files: print(k, d[k].shape, d[k].dtype)
