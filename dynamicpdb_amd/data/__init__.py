"""Host-side mirror of the reference's src/data diffusion classes (SE3Diffuser,
SO3Diffuser, R3Diffuser).  Table construction / sampling run in numpy on the host
exactly where the reference runs them (dataset workers, sampler); the score heads
that sit inside the model forward (calc_rot_score / calc_trans_score) and the
reverse step dispatch to the HIP kernels in dynamicpdb_amd.ops."""
from .se3_diffuser import SE3Diffuser  # noqa: F401
from .so3_diffuser import SO3Diffuser  # noqa: F401
from .r3_diffuser import R3Diffuser  # noqa: F401
