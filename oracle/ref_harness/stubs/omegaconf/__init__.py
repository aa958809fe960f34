"""Stub of omegaconf (test infrastructure)."""
class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e
    def __setattr__(self, k, v):
        self[k] = v
class OmegaConf:
    @staticmethod
    def to_yaml(c): return str(c)
    @staticmethod
    def to_container(c, resolve=True): return c
    @staticmethod
    def set_struct(c, f): return None
    @staticmethod
    def merge(*cs):
        out = DictConfig()
        for c in cs: out.update(c)
        return out
    @staticmethod
    def create(c=None): return DictConfig(c or {})
