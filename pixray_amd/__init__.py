"""MI355X-native pixray hot path.

hipGraph replay (engine.Session.enable_graph) needs the HIP runtime's graph *packet capture* off: with it on (the ROCm 7.2
default) a replayed graph keeps its kernels' argument blocks in memory that a later fresh `hipMalloc` can be handed, so
any new allocation between two replays may overwrite them (measured: tools/debug_capture.py).  The
runtime reads the flag once, at its first API call -- so it is set here, at package import, unless the caller chose a value;
a process that already touched the GPU before importing this package must export it itself."""
import os
import sys

GRAPH_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _runtime_already_started() -> bool:
    """True when this process has (possibly) initialised the HIP runtime before this import: the runtime reads the flag once,
    at its first API call, so a value set afterwards is not seen.  torch.cuda initialises lazily; `is_initialized()` is the
    public witness of that first call."""
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return True


# what enable_graph() relies on: the flag was in the environment BEFORE the runtime could read it -- either the caller
# exported it (any value: their choice is respected, and checked), or this import set it while the runtime was still cold
GRAPH_ENV_USER_SET = GRAPH_ENV in os.environ
GRAPH_ENV_IN_TIME = GRAPH_ENV_USER_SET or not _runtime_already_started()
os.environ.setdefault(GRAPH_ENV, "0")


def graph_replay_refusal():
    """None when hipGraph replay is safe in this process, else the reason (engine.Session.enable_graph reports it)."""
    if os.environ.get(GRAPH_ENV) != "0":
        return f"{GRAPH_ENV}=0 must be in the environment before the HIP runtime starts (it is {os.environ.get(GRAPH_ENV)!r})"
    if not GRAPH_ENV_IN_TIME:
        return (f"the HIP runtime was already initialised when pixray_amd was imported, so it never saw {GRAPH_ENV}=0: export it "
                f"before the first torch.cuda call (or import pixray_amd first)")
    return None
