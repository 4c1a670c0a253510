"""bsvd_amd -- MI355X-native engine for BSVD's streaming bidirectional-buffer forward path.

Importing the package registers ``BSVD`` in the arch registry (BasicSR's if importable), mirroring how
the reference fills ARCH_REGISTRY as an import side effect (Experimental_root/archs/__init__.py:5-9).
"""
from .registry import ARCH_REGISTRY, MODEL_REGISTRY, build_network  # noqa: F401
from .arch import BSVD  # noqa: F401
from .netspec import make_netspec  # noqa: F401
from .denoise import DenoisingModel, denoise_seq, temp_denoise  # noqa: F401
from .arch import TSN  # noqa: F401
from .pipeline import ClipPipeline  # noqa: F401

__version__ = "0.1.0"
