from _inert import Inert
def __getattr__(name):
    return Inert("open3d." + name)
