"""theanet_amd -- MI355X-native backend for theanet's convolutional training hot path.

``NeuralNet(layers, training_params, allwts)`` and the ``.prms`` layer-spec surface
are those of rakeshvar/theanet; the compute is hand-written HIP for gfx950 behind
the C-ABI in include/theanet_hip.h.  Importing the package is cheap; the HIP
library and the GPU are touched when the first NeuralNet / device array is made
(and that raises if either is missing -- there is no CPU fallback).
"""
from . import layer  # noqa: F401
from .device import DeviceArray, get_context, share  # noqa: F401
from .neuralnet import (NeuralNet, get_layers_info, get_training_params_info,  # noqa: F401
                        get_wts_info)

__version__ = "0.1.0"
