class Chain: pass
