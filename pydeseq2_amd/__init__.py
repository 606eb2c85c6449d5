"""pydeseq2_amd — MI355X (gfx950) engine for PyDESeq2's deseq2() -> Wald hot path.

* :class:`HipInference` — drop-in for ``pydeseq2.default_inference.DefaultInference``
  (the ``pydeseq2.inference.Inference`` plug-in interface, inference.py:9-362).
* :func:`deseq2` / :class:`DeseqPipeline` — the same path kept device-resident end to end.

GPU only: importing works anywhere, every computation requires libdeseq_hip.so and a GPU.
"""
from ._lib import Context, DsqError  # noqa: F401
from .inference import HipInference  # noqa: F401
from .pipeline import DeseqPipeline, deseq2  # noqa: F401
from .api import DeseqDataSet, DeseqStats  # noqa: F401  (AnnData-free facade with the reference's user-level names)
from . import summary  # noqa: F401  (module: summary.summary(res, contrast) = DeseqStats.summary() tail)

__version__ = "0.1.0"
