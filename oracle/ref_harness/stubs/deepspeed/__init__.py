"""Stub of deepspeed (test infrastructure)."""
class _U:
    @staticmethod
    def is_initialized():
        return False
utils = _U()
class _C:
    @staticmethod
    def is_configured():
        return False
    @staticmethod
    def checkpoint(fn, *a):
        return fn(*a)
checkpointing = _C()
