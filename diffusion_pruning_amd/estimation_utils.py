"""Relaxation utilities with the reference's names and numerics (pdm/utils/estimation_utils.py:5-75).

Semantics kept bit-for-bit where the reference is deterministic: the Gumbel noise is drawn on the HOST RNG exactly
as the reference does (``torch.rand`` on CPU; a fresh ``Generator().manual_seed(0)`` per call when ``fixed_seed``),
so seeded runs reproduce the reference's architecture codes.  Everything after the draw runs on the logits' device.
"""
from __future__ import annotations

import torch


def sample_gumbel(shape, eps: float = 1e-20, fixed_seed: bool = False) -> torch.Tensor:
    """G = -log(-log(U + eps) + eps), U ~ U(0,1) on the CPU generator (estimation_utils.py:5-10)."""
    gen = torch.Generator().manual_seed(0) if fixed_seed else None
    u = torch.rand(shape, generator=gen)
    return -torch.log(eps - torch.log(u + eps))


_GATHER_CACHE = {}

# Set by train_step.GraphedPrunerStep while the router is captured into / replayed from HIP graphs: the uniforms of every
# sample_gumbel_blocks call then live in STATIC device buffers that the host refills from its RNG stream before each replay.
NOISE_TAPE = None


class NoiseTape:
    """Host-RNG Gumbel noise for a CAPTURED router (reference: the noise is torch.rand on the CPU generator, quirk Q6,
    pdm/utils/estimation_utils.py:5-10).  A HIP graph cannot draw from the host generator, so every sample_gumbel_blocks call
    of the captured region is given a static device buffer for its uniforms (entries in call order).  ``refill()`` -- called
    on the host before each replay -- draws exactly what the eager calls would have drawn, in the same order from the same
    generator, and uploads it through the pinned ring; the captured kernels (gather into block order, -log(-log u)) then run on
    the new values.  Inside a stream capture no host draw happens (the buffers hold the warm-up pass's values)."""

    def __init__(self):
        self.entries = []         # [n, fixed_seed, key, device buffer]
        self.cursor = 0

    def begin(self):
        self.cursor = 0

    def uniforms(self, key, n, fixed_seed, device):
        i = self.cursor
        self.cursor += 1
        capturing = torch.cuda.is_current_stream_capturing()
        if i == len(self.entries):
            assert not capturing, "NoiseTape: a noise call first seen during capture (run the region eagerly once before capturing)"
            self.entries.append([n, bool(fixed_seed), key, torch.empty(n, dtype=torch.float32, device=device)])
        e = self.entries[i]
        assert e[0] == n and e[1] == bool(fixed_seed), "NoiseTape: the captured region's noise calls changed"
        if not capturing:
            self._fill(e)
        return e[3]

    @staticmethod
    def _fill(e):
        gen = torch.Generator().manual_seed(0) if e[1] else None
        u = torch.rand(e[0], generator=gen)
        e[3].copy_(_PinnedRing.of(("tape",) + tuple(e[2]), u.shape).send(u, e[3].device))

    def refill(self):
        """one host draw + upload per recorded call, in call order (the eager path's host-RNG consumption, exactly)"""
        for e in self.entries:
            self._fill(e)


def sample_gumbel_blocks(batch: int, widths, fixed_seed: bool = False, eps: float = 1e-20, device=None) -> torch.Tensor:
    """cat([sample_gumbel((batch, w)) for w in widths], dim=1), bit for bit, from ONE host draw.

    The CPU generator hands out its stream element by element, so k successive torch.rand calls equal one call of the total
    length (tests/test_host_logic.py pins that); a cached gather index puts the values where the per-block calls would have
    put them.  fixed_seed: every block restarts the seed-0 stream, i.e. block j holds the first batch*w_j values of it.
    ~430 host-side tensor ops per call in the per-block form, 5 here.  On a GPU the uniforms travel through pinned memory
    (a pageable copy would make the host wait for the stream)."""
    widths = tuple(int(w) for w in widths)
    key = (batch, widths, bool(fixed_seed))
    idx = _GATHER_CACHE.get(key)
    if idx is None:
        cols, off = [], 0
        for w in widths:
            cols.append(off + torch.arange(batch)[:, None] * w + torch.arange(w)[None, :])
            off += 0 if fixed_seed else batch * w
        idx = _GATHER_CACHE[key] = torch.cat(cols, dim=1)
    n = batch * (max(widths) if fixed_seed else sum(widths))
    if NOISE_TAPE is not None and device is not None and torch.device(device).type == "cuda":
        dkey = key + (str(device),)
        didx = _GATHER_CACHE.get(dkey)
        if didx is None:
            didx = _GATHER_CACHE[dkey] = idx.to(device)
        u = NOISE_TAPE.uniforms(key, n, fixed_seed, device)[didx]
        return -torch.log(eps - torch.log(u + eps))
    gen = torch.Generator().manual_seed(0) if fixed_seed else None
    u = torch.rand(n, generator=gen)
    if device is not None and torch.device(device).type == "cuda":
        # the uniforms are the host stream; the transform runs on the device (5 launches).  On the host it is a handful of
        # vectorised passes that ATen splits across OpenMP threads above 2,048 elements -- milliseconds each on a box whose
        # visible cores exceed its CPU share -- and the result differs from the host's only in the last ulp of two logs.
        dkey = key + (str(device),)
        didx = _GATHER_CACHE.get(dkey)
        if didx is None:
            didx = _GATHER_CACHE[dkey] = idx.to(device)
        u = _PinnedRing.of(key, u.shape).send(u, device)[didx]
    else:
        u = u[idx]
    g = -torch.log(eps - torch.log(u + eps))
    return g if device is None or g.device == torch.device(device) else g.to(device)


class _PinnedRing:
    """A few pinned staging buffers per noise shape, reused round-robin; an event per slot says when its last copy to the
    device has finished (allocating pinned memory per call costs milliseconds, a pageable copy blocks on the stream)."""
    SLOTS = 8
    _rings = {}

    @classmethod
    def of(cls, key, shape):
        r = cls._rings.get(key)
        if r is None:
            r = cls._rings[key] = cls(shape)
        return r

    def __init__(self, shape):
        self.buf = [torch.empty(shape, dtype=torch.float32).pin_memory() for _ in range(self.SLOTS)]
        self.ev = [None] * self.SLOTS
        self.i = 0

    def send(self, g, device):
        i = self.i
        self.i = (i + 1) % self.SLOTS
        if self.ev[i] is not None:
            self.ev[i].synchronize()
        self.buf[i].copy_(g)
        out = self.buf[i].to(device, non_blocking=True)
        ev = self.ev[i] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.ev[i] = ev
        return out


def hard_concrete(out: torch.Tensor) -> torch.Tensor:
    """Straight-through threshold at 0.5 (estimation_utils.py:67-75): value {0,1}, gradient identity."""
    hard = (out >= 0.5).to(out.dtype)
    return (hard - out).detach() + out


def _noisy_sigmoid(logits: torch.Tensor, temperature: float, offset, fixed_seed: bool) -> torch.Tensor:
    noise = sample_gumbel(logits.size(), fixed_seed=fixed_seed).to(logits.device)
    return torch.sigmoid((logits + noise + offset) / temperature)


def vector_gumbel_softmax(logits, temperature, offset=0, force_width_non_zero=False, fixed_seed=False):
    """estimation_utils.py:13-31: rows whose hard version is all-zero get +0.5 on their first entry."""
    y = _noisy_sigmoid(logits, temperature, offset, fixed_seed)
    if not force_width_non_zero:
        return y
    dead = hard_concrete(y).sum(dim=1) == 0
    if not bool(dead.any()):
        return y
    bumped = y.clone()
    bumped[dead, 0] = y[dead, 0] + 0.5
    return bumped


def gumbel_softmax_sample(logits, temperature, offset=0, force_width_non_zero=False, fixed_seed=False):
    """estimation_utils.py:34-46"""
    return vector_gumbel_softmax(logits, temperature, offset, force_width_non_zero, fixed_seed)


def importance_gumbel_softmax_sample(logits, temperature, offset=0, fixed_seed=False, noise=None):
    """estimation_utils.py:49-64: softmax -> cumsum -> flip gives monotone keep-probabilities; logit with eps=1e-6.
    noise: Gumbel noise already drawn (in the reference's order) and moved to the device by the caller."""
    p = torch.flip(torch.cumsum(torch.softmax(logits, dim=1), dim=1), dims=[1])
    eps = 1e-6
    x = torch.log(p + eps) - torch.log1p(-(p - eps))
    if noise is not None:
        return torch.sigmoid((x + noise + offset) / temperature)
    return _noisy_sigmoid(x, temperature, offset, fixed_seed)
