"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Makes the upstream reference (read-only, at /root/reference) importable inside
THIS container so that (a) golden vectors can be minted from the reference's own
code and (b) the oracle restatement in oracle/dfold_oracle.py can be pinned to it.
The reference does not exist on the GPU box: nothing under tests/ -m gpu,
bench.py or __graft_entry__.smoke() may import this module.

Recipe follows SURVEY.md Appendix A: a dozen stub packages for third-party
imports that the hot-path arithmetic never touches, plus sys.modules fakes for
torch.utils.tensorboard and src.analysis.*.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DFOLD_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "model"))


def install():
    """Put stubs + reference on sys.path (idempotent)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # reference tree is read-only
    for p in (REFERENCE_ROOT, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    if "torch.utils.tensorboard" not in sys.modules:
        tb = types.ModuleType("torch.utils.tensorboard")
        tb.SummaryWriter = object
        sys.modules["torch.utils.tensorboard"] = tb
    for name in ("src.analysis", "src.analysis.utils", "src.analysis.metrics"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    import src.analysis as _a  # noqa: F401  (fake)
    sys.modules["src.analysis"].utils = sys.modules["src.analysis.utils"]
    sys.modules["src.analysis"].metrics = sys.modules["src.analysis.metrics"]


class AttrDict(dict):
    """Plain attribute-dict standing in for an OmegaConf node."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def make_conf(frame_time: int, cache_dir: str = ".cache/"):
    """config/train_DFOLDv2.yaml + run_train.sh overrides as an attribute dict."""
    F = frame_time
    return AttrDict(
        diffuser=dict(
            dynamics=True, frame_time=F, diffuse_trans=True, diffuse_rot=True,
            r3=dict(min_b=0.1, max_b=20.0, coordinate_scaling=1.0),
            so3=dict(num_omega=1000, num_sigma=1000, min_sigma=0.1, max_sigma=1.5,
                     schedule="logarithmic", cache_dir=cache_dir, use_cached_score=False),
        ),
        model=dict(
            cfg_drop_rate=0.0, cfg_drop_in_train=True, cfg_gamma=2, frame_time=F,
            dirtect=False, dynamics=True, node_embed_size=256, edge_embed_size=128,
            dropout=0.0,
            embed=dict(DFOLDv2_embedder=True, index_embed_size=32, aatype_embed_size=32,
                       embed_self_conditioning=True, num_bins=22, min_bin=1e-5,
                       max_bin=20.0, skip_feature=False),
            ipa=dict(c_s=256, c_z=128, c_hidden=256, c_skip=64, no_heads=8,
                     no_qk_points=8, no_v_points=12, seq_tfmr_num_heads=4,
                     seq_tfmr_num_layers=2, num_blocks=4, coordinate_scaling=1.0,
                     spatial=True, temporal=False, temporal_position_encoding=True,
                     temporal_position_max_len=40, frozen_spatial=False),
        ),
        data=dict(dynamics=True, frame_time=F, min_t=0.01, num_t=10, is_extrapolation=False),
        experiment=dict(
            training=False, use_ddp=False, use_tensorboard=False, warm_start=None,
            ckpt_dir=None, eval_dir=None, learning_rate=1e-4, trans_loss_weight=100.0,
            rot_loss_weight=7.0, rot_loss_t_threshold=0.0, separate_rot_loss=False,
            torsion_loss_weight=1.0, bb_atom_loss_weight=1.0, bb_atom_loss_t_filter=0.25,
            dist_mat_loss_weight=1.0, dist_mat_loss_t_filter=0.25, aux_loss_weight=0.25,
            trans_x0_threshold=1.0, coordinate_scaling=1.0, noise_scale=1.0, name="probe",
            num_parameters=None, trainable_num_parameters=None),
    )
