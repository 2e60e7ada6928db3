"""Importable alias: ``import vln_ver_amd`` -> the package directory ``vln-ver_amd/``
(a hyphen cannot appear in an ``import`` statement)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module('vln-ver_amd')
sys.modules[__name__] = _pkg
