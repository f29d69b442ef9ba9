"""`from utils.muon_utils import Muon` (ref:train.py:326) -> the MI355X optimizer step.

`utils` is a namespace package in the reference (no __init__.py) and here too, so with this repository ahead of the
reference on PYTHONPATH the two `utils/` directories merge: this module wins for `utils.muon_utils`, while
`utils.mmrs_dataset`, `utils.data_utils`, ... (CPU data preparation, not replaced) still import from the reference.
"""
from cm3p_amd.muon import Muon  # noqa: F401

__all__ = ["Muon"]
