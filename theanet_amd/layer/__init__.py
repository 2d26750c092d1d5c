from .inlayers import InputLayer, ElasticLayer, ColorLayer, InputSlot
from .convpool import ConvLayer, PoolLayer, MeanLayer
from .hidden import HiddenLayer
from .dropout import DropOutLayer
from .outlayers import SoftmaxLayer, CenteredOutLayer, HingeLayer, ExpLossLayer, OutputLayer
from .auxiliary import SoftAuxLayer, AuxConcatLayer
from .layer import Layer, activation_by_name
