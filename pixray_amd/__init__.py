"""MI355X-native pixray hot path.

hipGraph replay (engine.Session.enable_graph) needs the HIP runtime's graph *packet capture* off: with it on (the ROCm 7.2
default) a replayed graph keeps its kernels' argument blocks in memory that a later fresh `hipMalloc` can be handed, so
any new allocation between two replays may overwrite them (measured: tools/debug_capture.py).  The
runtime reads the flag once, at its first API call -- so it is set here, at package import, unless the caller chose a value;
a process that already touched the GPU before importing this package must export it itself."""
import os

GRAPH_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
os.environ.setdefault(GRAPH_ENV, "0")
