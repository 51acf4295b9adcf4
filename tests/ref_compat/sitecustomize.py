"""Loaded by every Python process that has tests/ref_compat on PYTHONPATH: the worker processes the reference's
distributed tests spawn need the torchebm -> torchebm_amd import alias too (see ref_alias.py)."""
try:
    import ref_alias  # noqa: F401
except Exception:  # never break an interpreter start-up
    pass
