"""gtok — MI355X-native graph->sequence tokenizer (AGTT SENT trail walk + IBTT index serialiser).

The directory name carries a hyphen (it mirrors the upstream repository name), so import it with
``importlib.import_module("glearning-benchmark_amd")`` or, once that has happened, as ``gtok_amd``.
Putting this directory itself on ``sys.path`` exposes the drop-in ``graph_data_loader`` and
``autograph`` packages the reference's trainers import.
"""
import sys as _sys

from . import _lib, csr, dist, ops, synth  # noqa: F401
from ._lib import GtokError, build, lib  # noqa: F401
from .csr import GraphBatch  # noqa: F401

_sys.modules.setdefault("gtok_amd", _sys.modules[__name__])
