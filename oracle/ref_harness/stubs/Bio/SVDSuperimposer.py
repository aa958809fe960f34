class SVDSuperimposer: pass
