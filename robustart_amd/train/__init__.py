"""Solver entry points -- mirrors RobustART/train/__init__.py:1 (`cls_solver` re-export)."""
from . import cls_solver  # noqa: F401
