class HydraConfig:
    @staticmethod
    def initialized(): return False
    @staticmethod
    def get(): raise RuntimeError("stub")
