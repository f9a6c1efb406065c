"""HIP-graph helpers shared by the captured steps (train_step.py, pipeline.py) and the benchmark.

``new_graph()`` is ``torch.cuda.CUDAGraph()``; with ``KEEP_GRAPHS`` set (bench.py does, before capturing) the captured
hipGraph_t is kept next to its executable so that ``node_count`` can ask the runtime how many nodes -- kernel launches and
the few memcpy / memset nodes torch adds -- one replay issues.  Nothing here touches the device unless a graph exists."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

KEEP_GRAPHS = False
_hip = None


def new_graph() -> "torch.cuda.CUDAGraph":
    return torch.cuda.CUDAGraph(keep_graph=True) if KEEP_GRAPHS else torch.cuda.CUDAGraph()


def node_count(graph) -> Optional[int]:
    """nodes of a captured graph (None when the graph was not kept or the runtime cannot be asked)"""
    global _hip
    try:
        raw = graph.raw_cuda_graph()
    except Exception:  # noqa: BLE001  (not captured with keep_graph)
        return None
    try:
        if _hip is None:
            _hip = ctypes.CDLL("libamdhip64.so")
            _hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
            _hip.hipGraphGetNodes.restype = ctypes.c_int
        n = ctypes.c_size_t(0)
        rc = _hip.hipGraphGetNodes(ctypes.c_void_p(int(raw)), None, ctypes.byref(n))
        return int(n.value) if rc == 0 else None
    except Exception:  # noqa: BLE001
        return None
