"""Host-side mirror of `mogptk.gpr` for the MI355X exact-GP hot path."""
from .config import *
from .config import config
from .parameter import *
from .likelihood import *
from .kernel import *
from .singleoutput import *
from .multioutput import *
from .model import *
