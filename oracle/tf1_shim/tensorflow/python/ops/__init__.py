"""tf1_shim: package marker (test infrastructure, see ../__init__.py)."""
