"""Shared builders for the tests (the implementations live in the package: `iisan_amd/factory.py`)."""
from iisan_amd.factory import build_model, load_trainables, make_args  # noqa: F401
