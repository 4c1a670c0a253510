"""bsvd_amd -- MI355X-native engine for BSVD's streaming bidirectional-buffer forward path.

Importing the package registers ``BSVD_MI355X`` / ``TSN_MI355X`` / ``DenoisingModel_MI355X`` / ``ValFolderDataset_MI355X``
(BasicSR's registries if importable), mirroring how the reference fills its registries as an import side effect
(Experimental_root/archs/__init__.py:5-9); ``bsvd_amd.install(replace=True)`` puts the engine under the reference's stock
names.  Without BasicSR the package's own registries hold the stock names from the start.  See registry.py.
"""
from .registry import (ARCH_REGISTRY, DATASET_REGISTRY, MODEL_REGISTRY, build_dataset, build_model,  # noqa: F401
                       build_network, install, uninstall)
from .arch import BSVD  # noqa: F401
from .netspec import make_netspec  # noqa: F401
from .denoise import DenoisingModel, denoise_seq, temp_denoise  # noqa: F401
from .arch import TSN  # noqa: F401
from .pipeline import ClipPipeline  # noqa: F401
from .evaluation import ValFolderDataset  # noqa: F401

__version__ = "0.1.0"
