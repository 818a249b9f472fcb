from .linear_layers import *
from .seq_layers import *
from .image_layers import *
