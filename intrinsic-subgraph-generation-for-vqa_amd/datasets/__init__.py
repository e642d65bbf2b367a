"""Host-side data path in front of the hot path."""
from .gqa import gqa_collate  # noqa: F401
from .scene_graph import GQASceneGraphs  # noqa: F401
