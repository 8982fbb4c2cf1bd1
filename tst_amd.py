"""Import alias: the package directory is named ``text-style-transfer_amd`` (not a Python identifier),
so ``import tst_amd`` loads it through importlib and re-exports it under this name."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('text-style-transfer_amd')
sys.modules[__name__] = _pkg
