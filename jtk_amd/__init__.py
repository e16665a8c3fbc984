"""jtk_amd -- MI355X-native local clustering for JTK (ban-m/jtk haplotyper::local_clustering).

The product is the C-ABI shared library built from jtk_amd/csrc (include/jtk_lc.h); this package is the
thin Python harness around it used by tests/ and bench.py: flat batch layout, synthetic inputs and the
host-side mirror of the reference's `LocalClustering for DataSet` surface.
"""
from . import ffi  # noqa: F401
from .batch import Batch, default_params  # noqa: F401

__all__ = ["ffi", "Batch", "default_params"]
