"""model.encoder.encoders.psp_encoders (reference: psp_encoders.py:35-116; built by util.py:142-160)."""
from vtoonify_amd.psp import GradualStyleEncoder  # noqa: F401

__all__ = ["GradualStyleEncoder"]
