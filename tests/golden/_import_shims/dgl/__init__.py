from _inert import Inert
from . import nn, function, utils, geometry  # noqa
def __getattr__(name):
    return Inert("dgl." + name)
