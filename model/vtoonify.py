"""model.vtoonify (reference: model/vtoonify.py:130-286) -> the gfx950 implementation."""
from vtoonify_amd.vtoonify import VToonify  # noqa: F401

__all__ = ["VToonify"]
