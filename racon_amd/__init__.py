"""racon_amd — MI355X-native window-consensus engine behind racon's Polisher/Window surface.

Only the hot path of lbcb-sci/racon lives here (reference src/window.cpp:65-149
over spoa); see DESIGN.md.  The HIP engine is loaded lazily from
racon_amd/csrc/libracon_hip.so and there is NO CPU fallback: importing
`racon_amd.engine` without the built library (or without a GPU) raises.
"""
from .batch import WindowBatch, ConsensusResult  # noqa: F401

__version__ = "0.1.0"
