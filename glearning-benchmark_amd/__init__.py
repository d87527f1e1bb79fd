"""gtok — MI355X-native graph->sequence tokenizer (AGTT SENT trail walk + IBTT index serialiser).

The directory name carries a hyphen (it mirrors the upstream repository name), so import it with
``importlib.import_module("glearning-benchmark_amd")``.  Putting this directory itself on ``sys.path``
exposes the drop-in ``graph_data_loader`` and ``autograph`` packages the reference's trainers import
(INTEGRATION.md).
"""
from . import _lib  # noqa: F401
from ._lib import GtokError, build, lib  # noqa: F401
from . import csr  # noqa: F401
from .csr import GraphBatch  # noqa: F401
from . import ops, dist, synth, rows  # noqa: F401,E401
from . import torch_ops  # noqa: F401  (registers torch.ops.gtok.*)
from . import graph_data_loader  # noqa: F401  (needs ops / GraphBatch above)
from . import agtt  # noqa: F401
from .autograph.datamodules.data.tokenizer import Graph2TrailTokenizer  # noqa: F401
