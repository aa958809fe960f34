class PDBParser: pass
class PDBIO: pass
class MMCIFParser: pass
