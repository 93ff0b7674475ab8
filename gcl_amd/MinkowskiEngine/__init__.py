from . import utils
