"""Checkpoint writer / warm start / true resume of the training step.

Reference: `write_checkpoint` (src/data/utils.py:324-362: one torch-pickled dict with the keys model / conf / optimizer /
epoch / step, called from train_DFOLD_dynamics.py:736-758 as `step_{trained_steps}.pth`) and the warm start
`Experiment.load_pretrianed_model` (train_DFOLD_dynamics.py:468-499: 'module.' prefixes stripped, tensors whose name or
shape does not match are skipped, optimizer / epoch / step NOT restored -- the reference cannot resume a run).

Same file format in both directions (a reference checkpoint loads here, a checkpoint written here loads in the
reference), plus `resume`, which also restores the optimizer (FusedAdam keeps torch.optim.Adam's state layout, so the
states are interchangeable), the epoch / step counters and the device RNG position: a resumed run continues bit-for-bit."""
import copy
import os

import torch


def write_checkpoint(ckpt_path, model, conf, optimizer, epoch, step, logger=None, use_torch=True, extra=None):
    """Signature and file content of the reference's data.utils.write_checkpoint: `model` / `optimizer` are STATE DICTS
    (the reference passes deep copies, train:746-750).  Written to a temporary file and renamed, so that a crash never
    leaves a truncated checkpoint behind.  `extra`: optional dict stored under 'extra' (e.g. the device RNG position)."""
    if not use_torch:
        raise ValueError("only the torch serialisation (use_torch=True, as the reference's call site) is supported")
    msg = f'Serializing experiment state to {ckpt_path}'
    logger.info(msg) if logger is not None else print(msg)
    os.makedirs(os.path.dirname(os.path.abspath(ckpt_path)), exist_ok=True)
    payload = {'model': model, 'conf': conf, 'optimizer': optimizer, 'epoch': epoch, 'step': step}
    if extra is not None:
        payload['extra'] = extra
    tmp = f"{ckpt_path}.tmp.{os.getpid()}"
    torch.save(payload, tmp, pickle_protocol=4)
    os.replace(tmp, ckpt_path)


class _AllowListedPickle:
    """`pickle_module` for torch.load: the standard unpickler (every protocol -- the reference writes protocol 4, which
    torch's own weights_only unpickler does not read) with `find_class` restricted to what a checkpoint of this path is
    made of: tensors and their storages, plain containers, numpy scalars, OmegaConf's container / node classes (the
    reference's 'conf', src/data/utils.py:353-362).  Anything else raises instead of being imported and called."""
    import pickle as _pickle

    ALLOWED = {
        ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "set"), ("builtins", "frozenset"),
        ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"),
        ("builtins", "str"), ("builtins", "bool"), ("builtins", "complex"), ("builtins", "bytes"), ("builtins", "slice"),
        ("typing", "Any"), ("typing", "Union"), ("typing", "Optional"), ("typing", "Dict"), ("typing", "List"),
        ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_tensor"),
        ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
        ("torch.nn.parameter", "Parameter"), ("torch.serialization", "_get_layout"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "dtype"),
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
    }

    # OmegaConf's container / node / metadata classes by NAME (a pickled DictConfig is made of these and plain containers).
    # An explicit set, not a pattern: protocol 4 resolves dotted names attribute by attribute, so a rule like "any capitalised
    # name under omegaconf" lets 'Marker.__init__.__globals__.get' through and from there to any callable.
    OMEGACONF = {
        "DictConfig", "ListConfig", "ContainerMetadata", "Metadata", "AnyNode", "StringNode", "IntegerNode", "FloatNode",
        "BooleanNode", "EnumNode", "BytesNode", "PathNode", "UnionNode", "InterpolationResultNode", "ValueNode", "Node",
        "Container", "Box", "BaseContainer", "SCMode", "Flags",
    }

    class Unpickler(_pickle.Unpickler):
        def find_class(self, module, name):
            omega = module.split(".")[0] == "omegaconf" and name in _AllowListedPickle.OMEGACONF
            ok = "." not in name and (
                (module, name) in _AllowListedPickle.ALLOWED
                or (module == "torch" and (name.endswith("Storage") or name in _AllowListedPickle._torch_dtypes()))
                or omega)
            if not ok:
                raise _AllowListedPickle._pickle.UnpicklingError(
                    f"global {module}.{name} is not on the checkpoint allow-list (dynamicpdb_amd/checkpoint.py)")
            obj = super().find_class(module, name)
            if omega and not isinstance(obj, type):
                raise _AllowListedPickle._pickle.UnpicklingError(f"global {module}.{name} is not a class")
            return obj

    @staticmethod
    def _torch_dtypes():
        return {n for n in dir(torch) if isinstance(getattr(torch, n), torch.dtype)}

    @classmethod
    def load(cls, f, **kw):
        return cls.Unpickler(f, **kw).load()

    Pickler = _pickle.Pickler
    dump, dumps = _pickle.dump, _pickle.dumps
    __name__ = "dynamicpdb_amd.checkpoint._AllowListedPickle"


def read_checkpoint(ckpt_path, allow_pickle=None):
    """Loads a checkpoint WITHOUT executing anything from the file: an allow-listed unpickler (`_AllowListedPickle`:
    tensors, containers, numbers, OmegaConf nodes; every pickle protocol -- the reference and `write_checkpoint` use
    protocol 4).  The unrestricted pickle loader (arbitrary code execution from the file) is used only on request --
    `allow_pickle=True`, or DFOLD_TRUSTED_CHECKPOINTS=1 in the environment (default: off) -- and says so with a warning
    that names the file."""
    import pickle
    import warnings
    try:
        return torch.load(ckpt_path, map_location='cpu', weights_only=False, pickle_module=_AllowListedPickle)
    except (pickle.UnpicklingError, RuntimeError, AttributeError, TypeError, ImportError) as e:
        if allow_pickle is None:
            allow_pickle = os.environ.get("DFOLD_TRUSTED_CHECKPOINTS", "0") == "1"
        if not allow_pickle:
            raise RuntimeError(f"{ckpt_path} needs the unrestricted pickle loader ({e}); pass allow_pickle=True or set "
                               "DFOLD_TRUSTED_CHECKPOINTS=1 for files you trust") from e
        warnings.warn(f"{ckpt_path}: falling back to the UNRESTRICTED pickle loader (code in the file is executed)")
        return torch.load(ckpt_path, map_location='cpu', weights_only=False)


def load_pretrained_model(model, ckpt_path, logger=None):
    """Warm start as Experiment.load_pretrianed_model: returns True on success; parameters missing from the checkpoint or
    of a different shape keep their current values."""
    log = logger.info if logger is not None else print
    err = logger.error if logger is not None else print
    try:
        log(f'Loading checkpoint from {ckpt_path}')
        ckpt = read_checkpoint(ckpt_path)
        if ckpt is None or 'model' not in ckpt:
            err("Checkpoint or model not found in checkpoint file.")
            return False
        ckpt_model = ckpt['model']
        if ckpt_model is None:
            err("Checkpoint model is None.")
            return False
        ckpt_model = {k.replace('module.', ''): v for k, v in ckpt_model.items()}
        sd = model.state_dict()
        sd.update({k: v for k, v in ckpt_model.items() if k in sd and v.shape == sd[k].shape})
        model.load_state_dict(sd)
        log(f'Warm starting from: {ckpt_path}')
        return True
    except Exception as e:  # the reference logs and carries on
        err(f"Error loading checkpoint: {e}")
        return False


def save(trainer, ckpt_path, conf=None, epoch=0, step=0, rng=None, logger=None):
    """checkpoint of a dynamicpdb_amd.experiment.Trainer in the reference's format (state dicts moved to the host)"""
    cpu = lambda o: (o.detach().cpu() if torch.is_tensor(o) else {k: cpu(v) for k, v in o.items()} if isinstance(o, dict)
                     else type(o)(cpu(v) for v in o) if isinstance(o, (list, tuple)) else copy.deepcopy(o))
    write_checkpoint(ckpt_path, cpu(trainer.model.state_dict()), conf, cpu(trainer.opt.state_dict()), epoch, step,
                     logger=logger, extra=None if rng is None else {'rng': rng.state()})


def resume(trainer, ckpt_path, rng=None, strict=True):
    """True resume: model (strict), optimizer state (exp_avg / exp_avg_sq / max_exp_avg_sq / step of every parameter), the
    epoch / step counters and -- if the checkpoint holds one -- the device RNG position.  Returns (epoch, step, conf)."""
    ckpt = read_checkpoint(ckpt_path)
    model_sd = {k.replace('module.', ''): v for k, v in ckpt['model'].items()}
    trainer.model.load_state_dict(model_sd, strict=strict)
    if ckpt.get('optimizer') is not None:
        trainer.opt.load_state_dict(ckpt['optimizer'])
    if rng is not None and ckpt.get('extra') and 'rng' in ckpt['extra']:
        rng.seed, rng.subseq = ckpt['extra']['rng']['seed'], ckpt['extra']['rng']['subseq']
    return ckpt.get('epoch', 0), ckpt.get('step', 0), ckpt.get('conf')
