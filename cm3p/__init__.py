"""Drop-in `cm3p` package: put this repository ahead of the reference checkout on PYTHONPATH and the reference's
`train.py` (`from cm3p import CM3PModel, CM3PConfig`, ref:train.py:14-18) gets the MI355X implementation of the
modeling / configuration modules, while every other `cm3p.*` module (processing, tokenization, parsing: CPU data
preparation, out of scope here) still resolves to the reference's own files further down the path.
"""
import os
import sys

__path__ = [os.path.dirname(os.path.abspath(__file__))]
for _entry in sys.path:  # namespace-style extension: later `cm3p/` directories serve the modules we do not replace
    _cand = os.path.join(_entry or ".", "cm3p")
    if os.path.isdir(_cand) and os.path.abspath(_cand) not in [os.path.abspath(p) for p in __path__]:
        __path__.append(_cand)

from .configuration_cm3p import *  # noqa: F401,F403,E402
from .modeling_cm3p import *  # noqa: F401,F403,E402
