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

out = os.path.join(os.path.dirname(__file__), "..", "..", "dynamicpdb_amd", "data", "residue_tables.npz")
np.savez_compressed(
    out,
    default_frames=np.asarray(rc_src.restype_rigid_group_default_frame, np.float32),      # [21,8,4,4]
    atom14_group=np.asarray(rc_src.restype_atom14_to_rigid_group, np.int64),             # [21,14]
    atom14_mask=np.asarray(rc_src.restype_atom14_mask, np.float32),                      # [21,14]
    atom14_pos=np.asarray(rc_src.restype_atom14_rigid_group_positions, np.float32),      # [21,14,3]
    atom37_to_atom14=np.asarray(rc_of.RESTYPE_ATOM37_TO_ATOM14, np.int64),               # [21,37]
    atom37_mask=np.asarray(rc_of.RESTYPE_ATOM37_MASK, np.float32),                       # [21,37]
)
d = np.load(out)
for k in d.files: print(k, d[k].shape, d[k].dtype)
