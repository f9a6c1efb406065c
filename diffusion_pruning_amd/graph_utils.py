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


def concurrent_stream(main: Optional["torch.cuda.Stream"] = None, tries: int = 12, spin_ms: float = 0.4, log=None) -> "torch.cuda.Stream":
    """A side stream whose work really runs NEXT TO `main` (default: the current stream).

    Round 4 traced the "capture-order variance" of the training steps (the same captured step replaying 6-10 % faster or
    slower depending on when in the process it was captured) to this: the three graphs of a step take the
    same time whichever capture they belong to, but the teacher graph on the side stream overlaps the student's forward only
    for SOME side streams -- torch hands out streams from a pool, the runtime maps them onto a few hardware queues, and a side
    stream that shares its queue with the launching stream serialises behind it.  So the side stream is chosen by measurement:
    candidates are tried with two spin kernels, one per stream, until their total time says they overlapped."""
    main = main or torch.cuda.current_stream()
    dev = main.device
    cycles = int(spin_ms * 1e-3 * 2.0e9)

    def timed(side):
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            e0.record(main)
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    torch.cuda._sleep(cycles)
            torch.cuda._sleep(cycles)
            if side is not None:
                main.wait_stream(side)
            e1.record(main)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1)
    timed(None)
    single = min(timed(None), timed(None))
    best, best_ratio, seen = None, 1e9, []
    for _ in range(tries):
        s = torch.cuda.Stream(device=dev)
        timed(s)
        r = min(timed(s), timed(s)) / single
        seen.append(round(r, 2))
        if r < best_ratio:
            best, best_ratio = s, r
        if r < 1.3:
            break
    if log is not None:
        log.append({"single_ms": round(single, 3), "ratios_tried": seen, "chosen_ratio": round(best_ratio, 2)})
    return best
