"""tpgan_amd -- MI355X-native neighbourhood ops + set-abstraction path of TPU-GAN.

Directory name on disk: ``temporal-pointcloud-upsampling-gan_amd`` (import it as
``tpgan_amd`` through the alias module at the repo root).

Public surface
  ops                      functional ops (HIP through the C-ABI, autograd glue)
  compat.install()         puts import-compatible ``pointnet2_ops``, ``pytorch3d``,
                           ``frnn`` and ``chamferdist`` modules on sys.path so that the
                           reference's gcn_lib / discriminator.py / loss.py run unchanged
"""
import os as _os
import sys as _sys

__version__ = "0.1.0"

COMPAT_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "compat")


def install_compat():
    """Make `import pointnet2_ops`, `pytorch3d.ops`, `frnn`, `chamferdist` resolve to this package."""
    if COMPAT_DIR not in _sys.path:
        _sys.path.insert(0, COMPAT_DIR)
    return COMPAT_DIR


from . import ops  # noqa: E402,F401
