"""Stub of biopython (test infrastructure)."""
