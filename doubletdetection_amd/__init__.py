"""MI355X-native drop-in for the hot path of DoubletDetection: ``BoostClassifier``.

``from doubletdetection_amd import BoostClassifier`` replaces
``from doubletdetection import BoostClassifier``; the boosting loop runs in hand-written HIP kernels
(libddx.so, C-ABI in include/ddx.h).  Importing the package does not touch the GPU; constructing the
device context inside ``fit`` fails loudly when libddx.so or a gfx950 device is missing.
"""
from .classifier import BoostClassifier, release_device_memory

__version__ = "0.1.0"
__all__ = ["BoostClassifier", "release_device_memory"]
