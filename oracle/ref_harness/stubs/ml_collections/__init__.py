"""Stub of ml_collections (test infrastructure): attribute dict + FieldReference."""
class FieldReference:
    def __init__(self, v, field_type=None):
        self._v = v
    def get(self):
        return self._v
    def _bin(self, other):
        return self
    __mul__ = __rmul__ = __add__ = __radd__ = __floordiv__ = __truediv__ = __sub__ = _bin

class ConfigDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v
    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            v = ConfigDict(v)
        if isinstance(v, FieldReference):
            v = v.get()
        super().__setitem__(k, v)
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e
    def __setattr__(self, k, v):
        self[k] = v
    def copy_and_resolve_references(self):
        import copy
        return copy.deepcopy(self)
    def unlocked(self):
        import contextlib
        return contextlib.nullcontext(self)
    def lock(self):
        return self

def placeholder(*a, **k):
    return None
