"""Resolve the ONE canonical root package (glearning-benchmark_amd) no matter how this subpackage was
imported: as glearning-benchmark_amd.<sub> or, in the drop-in layout, as a top-level package with
glearning-benchmark_amd/ itself on sys.path.  A second copy of the root would duplicate the ctypes types."""
import importlib
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def root():
    name = os.path.basename(_PKG_DIR)
    if name in sys.modules:
        return sys.modules[name]
    parent = os.path.dirname(_PKG_DIR)
    if parent not in sys.path:
        sys.path.insert(0, parent)
    return importlib.import_module(name)
