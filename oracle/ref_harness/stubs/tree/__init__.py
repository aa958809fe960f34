"""Stub of dm-tree: only map_structure over dict/list/tuple (test infrastructure)."""
def map_structure(fn, *s):
    x = s[0]
    if isinstance(x, dict):
        return {k: map_structure(fn, *[t[k] for t in s]) for k in x}
    if isinstance(x, (list, tuple)):
        out = [map_structure(fn, *[t[i] for t in s]) for i in range(len(x))]
        return type(x)(out) if not hasattr(x, "_fields") else type(x)(*out)
    return fn(*s)
