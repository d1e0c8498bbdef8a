"""model.bisenet.model (reference: model/bisenet/model.py:215-254; used at style_transfer.py:66-68,171)."""
from vtoonify_amd.bisenet import BiSeNet  # noqa: F401

__all__ = ["BiSeNet"]
