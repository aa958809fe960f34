class rms: pass
class align: pass
class rdf: pass
class contacts: pass
