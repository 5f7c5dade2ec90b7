"""depthhead_amd -- MI355X-native Hough-forest head-pose inference (the
`HoughPrediction::predict_parameter_parallel` path of Entscheider/depthhead)."""
from .forest import Forest, NODE_DTYPE  # noqa: F401
from .synth import ModelParams  # noqa: F401

__all__ = ["Forest", "NODE_DTYPE", "ModelParams"]
