"""Import alias: ``import tpgan_amd`` loads the package that lives in the
directory ``temporal-pointcloud-upsampling-gan_amd/`` (the hyphens in the
mandated directory name are not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "temporal-pointcloud-upsampling-gan_amd")
_spec = importlib.util.spec_from_file_location(
    "tpgan_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tpgan_amd"] = _mod
_spec.loader.exec_module(_mod)
