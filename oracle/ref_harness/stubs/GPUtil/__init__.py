def getAvailable(order='memory', limit=8, **k): return []
