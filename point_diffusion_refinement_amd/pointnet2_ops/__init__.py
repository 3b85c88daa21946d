"""Drop-in for the reference's `pointnet2_ops` package (pointnet2_ops/__init__.py:1-3)."""
from . import pointnet2_modules, pointnet2_utils  # noqa: F401
