from .inlayers import InputLayer, ElasticLayer, ColorLayer, InputSlot
from .convpool import ConvLayer, PoolLayer, MeanLayer
from .hidden import HiddenLayer
from .dropout import DropOutLayer
from .outlayers import SoftmaxLayer, CenteredOutLayer, HingeLayer, ExpLossLayer, OutputLayer
from .layer import Layer, activation_by_name

# Reference layer types outside the accelerated hot path: naming them
# keeps NeuralNet's getattr(layer, name) lookup giving a clear error instead of AttributeError.
_OUT_OF_SCOPE = ("SoftAuxLayer", "AuxConcatLayer")


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise NotImplementedError(
            "Unknown Layer Type" + name + " (reference layer outside the MI355X hot path)")
    raise AttributeError(name)
