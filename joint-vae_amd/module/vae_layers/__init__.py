from .layers import Encoder, Sampling, Classifier, Sigma
from .conv import build_de_conv_layers, find_input_shape
from .misc import activation_layers, onehot_encoding
