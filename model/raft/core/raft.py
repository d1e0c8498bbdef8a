"""model.raft.core.raft (reference: model/raft/core/raft.py:24-144; used at smooth_parsing_map.py:12,97-102,154)."""
from vtoonify_amd.raft import RAFT  # noqa: F401

__all__ = ["RAFT"]
