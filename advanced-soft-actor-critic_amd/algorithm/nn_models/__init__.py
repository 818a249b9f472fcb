"""Model-plugin surface (`import algorithm.nn_models as m` in user `envs/*/nn*.py` files)."""
from .layers import *
from .representation import *
from .q import *
from .policy import *
from .exploration import *
from .predictions import *
