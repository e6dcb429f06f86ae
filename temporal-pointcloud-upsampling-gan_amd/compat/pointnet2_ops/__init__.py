"""Import-compatible stand-in for erikwijmans' ``pointnet2_ops`` (README.md:8 of the reference)."""
from . import pointnet2_utils  # noqa: F401
