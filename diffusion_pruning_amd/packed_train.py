"""Packed-parameter training of a pruned expert (BASELINE configs[4]; the counterpart of FineTuner's
optimizer state, pdm/training/trainer.py:1529-1540).

MI355X-first: during fine-tuning the trainable state lives in the KERNELS' layout.  For every contraction of the expert the
optimizer owns one fp32 tensor in the packed order ``[N][taps][cin_pad]`` -- compact: only the live rows / columns of the
architecture code -- and the bf16 operand the kernels read is its shadow:

  * the weight-gradient kernel (aptp_conv_wgrad) already produces ``[N][taps][C]``: the gradient needs no scatter into a
    full-shape OIHW tensor (three full-size passes per weight in the diffusers layout);
  * after ``optimizer.step()`` the shadows are refreshed with ONE multi-tensor cast (``torch._foreach_copy_``) plus one
    strided copy per data-gradient operand (the 180-degree-rotated transpose), instead of re-packing every weight from a
    diffusers-layout master (gather + permute + cast + pad: ~10 launches per weight, ~7,000 per step);
  * dead channels / heads / FF chunks / dropped blocks carry no optimizer state at all;
  * nothing in the step depends on the host, so forward + backward replay from HIP graphs and the optimizer + refresh follow as three launches
    (train_step.GraphedFineTunerStep).

Biases and norm affine parameters are compact fp32 tensors that the kernels read directly (no shadow).  The diffusers-named
full-shape parameters stay the checkpoint interface: ``export_()`` writes the packed values back into them.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import autograd as AG
from . import ops


class _Gemm:
    """one contraction: packed fp32 master ``P`` [N, taps, cin_pad], optional compact bias ``Pb`` [N] (aliased by the pack),
    bf16 shadow ``pw`` (forward operand) and ``pwb`` (data-gradient operand, built on first use)"""

    def __init__(self, weight, bias: Optional[nn.Parameter], pw: ops.PackedWeight,
                 lo: Optional[torch.Tensor], li: Optional[torch.Tensor]):
        assert not pw.geglu and pw.ln_colsum is None and pw.Cin2 == 0, "packed training uses the plain packs of the ft path"
        # `weight` may be a tuple of equally shaped parameters that share the input (to_q | to_k | to_v of an attention block):
        # ONE contraction whose output rows are the parts' live rows back to back; `lo` / `li` apply to every part
        self.weights = tuple(weight) if isinstance(weight, (tuple, list)) else (weight,)
        assert bias is None or len(self.weights) == 1
        self.weight, self.bias, self.pw, self.lo, self.li = self.weights[0], bias, pw, lo, li
        dev = pw.w.device
        ws = []
        for wp in self.weights:
            w = wp.detach().to(device=dev, dtype=torch.float32)
            if w.dim() == 2:
                w = w[:, :, None, None]
            if lo is not None:
                w = w[lo.to(dev)]
            if li is not None:
                w = w[:, li.to(dev)]
            ws.append(w)
        self.n_part = ws[0].shape[0]
        w = torch.cat(ws, 0) if len(ws) > 1 else ws[0]
        self.n_live, self.c_live, self.KH, self.KW = w.shape
        assert self.n_live <= pw.N and self.c_live <= pw.Cin and (self.KH, self.KW) == (pw.KH, pw.KW)
        P = torch.zeros(pw.w.shape, dtype=torch.float32, device=dev)
        P[:self.n_live, :, :self.c_live] = w.permute(0, 2, 3, 1).reshape(self.n_live, self.KH * self.KW, self.c_live)
        self.P = nn.Parameter(P)
        self.Pb = None
        if pw.bias is not None:
            # the pack's fp32 bias becomes the trainable tensor, read directly by the kernels (a fresh tensor: an un-gathered
            # pack bias can alias the diffusers-layout master, whose storage and version counter must stay untouched)
            self.Pb = nn.Parameter(pw.bias.detach().clone())
            pw.bias = self.Pb.data
        pw.w.copy_(P)                                  # shadow = bf16(master), exactly what a re-pack would give
        self.pwb: Optional[ops.PackedWeight] = None

    def get_bwd(self) -> ops.PackedWeight:
        if self.pwb is None:
            w4 = self.P.detach()[:, :, :self.pw.Cin].reshape(self.pw.N, self.KH, self.KW, self.pw.Cin).permute(0, 3, 1, 2)
            self.pwb = ops.pack_weight_dgrad(w4, device=self.P.device)
        return self.pwb

    def refresh_bwd_(self):
        if self.pwb is not None:
            ops.pack_dgrad_from_packed(self.pw, self.pwb)          # one tiled transpose (was: flip + strided copy)

    def parts(self):
        """(parameter, row range of P / P.grad it owns) for every weight of the contraction"""
        return [(wp, slice(i * self.n_part, (i + 1) * self.n_part)) for i, wp in enumerate(self.weights)]

    @torch.no_grad()
    def export_(self):
        for wp, rows in self.parts():
            g = self.P.detach()[rows, :, :self.c_live].permute(0, 2, 1).reshape(self.n_part, self.c_live, self.KH, self.KW)
            full = wp.data
            g = g.to(device=full.device, dtype=full.dtype)
            if full.dim() == 2:
                g = g.reshape(self.n_part, self.c_live)
            ro = self.lo.to(full.device) if self.lo is not None else torch.arange(full.shape[0], device=full.device)
            ci = self.li.to(full.device) if self.li is not None else torch.arange(full.shape[1], device=full.device)
            full[ro[:, None], ci[None, :]] = g
        if self.bias is not None and self.Pb is not None:
            b = self.Pb.detach()[:self.n_live].to(device=self.bias.device, dtype=self.bias.dtype)
            if self.lo is not None:
                self.bias.data[self.lo.to(self.bias.device)] = b
            else:
                self.bias.data.copy_(b)


class _Affine:
    """compact fp32 (gamma, beta) of a GroupNorm / LayerNorm, read directly by the kernels"""

    def __init__(self, gamma_p: nn.Parameter, beta_p: nn.Parameter, gamma: torch.Tensor, beta: torch.Tensor,
                 live: Optional[torch.Tensor]):
        self.gamma_p, self.beta_p, self.live = gamma_p, beta_p, live
        self.Pg, self.Pb = nn.Parameter(gamma.detach().float().clone()), nn.Parameter(beta.detach().float().clone())

    @torch.no_grad()
    def export_(self):
        for P, full in ((self.Pg, self.gamma_p), (self.Pb, self.beta_p)):
            v = P.detach().to(device=full.device, dtype=full.dtype)
            if self.live is not None:
                full.data[self.live.to(full.device)] = v
            else:
                full.data.copy_(v)


class PackedTrainer:
    """Registry of the packed trainable state of one expert.  ``attach`` switches the model's fine-tuning forward to it;
    entries are created the first time a module runs (call ``materialize`` with one batch before building the optimizer)."""

    def __init__(self, model):
        self.model = model
        self.gemms: Dict[int, _Gemm] = {}
        self.affines: Dict[int, _Affine] = {}

    def attach(self):
        for m in self.model.modules():
            m.__dict__["_pk"] = self
        return self

    def detach(self):
        for m in self.model.modules():
            m.__dict__.pop("_pk", None)

    # ---- called by the model's fine-tuning forward in place of AG.conv_w / AG.GroupNormWFn / AG.LayerNormWFn -----------------
    def conv(self, x, weight, bias, pw, get_bwd, stride=1, pad=None, ups=0, out_f32=False, live_out=None, live_in=None,
             residual=None, rowbias=None):
        key = id(weight[0]) if isinstance(weight, (tuple, list)) else id(weight)
        e = self.gemms.get(key)
        if e is None:
            e = self.gemms[key] = _Gemm(weight, bias, pw, live_out, live_in)
        assert e.pw is pw, "the plan's packs were rebuilt under a packed trainer (call PackedTrainer after the last set_structure)"
        return AG.conv_p(x, e.P, e.Pb, e.pw, e.get_bwd, stride=stride, pad=pad, ups=ups, out_f32=out_f32, residual=residual,
                         rowbias=rowbias)

    def groupnorm(self, x, gamma_p, beta_p, gamma, beta, groups, eps, silu, C, live, fork=False):
        a = self.affines.get(id(gamma_p))
        if a is None:
            a = self.affines[id(gamma_p)] = _Affine(gamma_p, beta_p, gamma, beta, live)
        return AG.GroupNormWFn.apply(x, a.Pg, a.Pb, a.Pg, a.Pb, groups, eps, silu, C, None, fork)

    def layernorm(self, x, gamma_p, beta_p, gamma, beta, eps, fork=False):
        a = self.affines.get(id(gamma_p))
        if a is None:
            a = self.affines[id(gamma_p)] = _Affine(gamma_p, beta_p, gamma, beta, None)
        return AG.LayerNormWFn.apply(x, a.Pg, a.Pb, a.Pg, a.Pb, eps, fork)

    # ---- optimizer-facing API ---------------------------------------------------------------------------------------------------
    def materialize(self, *forward_args, **forward_kwargs):
        """one forward + backward (on a zero loss scale) so that every entry, including the data-gradient operands, exists.
        Runs on a side stream: autograd anchors a parameter's AccumulateGrad node to the stream of its first backward and
        re-uses the node while any graph that reached it is alive; a node anchored to the legacy default stream cannot take
        part in a later HIP-graph capture (hipStreamEndCapture dies on the cross-stream edge)."""
        if torch.cuda.is_available():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = self.model(*forward_args, **forward_kwargs).sample
                (out.float().sum() * 0.0).backward()
                del out
            torch.cuda.current_stream().wait_stream(side)
        else:
            out = self.model(*forward_args, **forward_kwargs).sample
            (out.float().sum() * 0.0).backward()
            del out
        for p in self.parameters():
            p.grad = None
        return self

    def parameters(self) -> List[nn.Parameter]:
        ps: List[nn.Parameter] = []
        for e in self.gemms.values():
            ps.append(e.P)
            if e.Pb is not None:
                ps.append(e.Pb)
        for a in self.affines.values():
            ps += [a.Pg, a.Pb]
        return ps

    @torch.no_grad()
    def ensure_grad_buffers(self, arena: bool = False, world: int = 1):
        """persistent, zero-initialised .grad for every packed tensor (direct-gradient mode, ops.GRAD_DIRECT): the kernels write
        the live region of each buffer every step and never touch the pad columns.
        arena=True: all of them are views into ONE contiguous fp32 buffer (``self.grad_arena``; ``self.grad_offsets`` = element
        offset per parameter, in self.parameters() order), laid out in REVERSE registration order -- the order a backward
        produces gradients in -- with 256-byte aligned views; slot 0 (the first 64 elements) is reserved for the step's loss
        (train_step.ArenaGradReducer: the data-parallel exchange then runs in place on contiguous ranges, and its sum of the losses
        is the ranks' common NaN verdict); the total is padded to a multiple of 64 * world elements (equal shards)."""
        ps = self.parameters()
        if not arena:
            for p in ps:
                if p.grad is None or p.grad.shape != p.shape:
                    p.grad = torch.zeros_like(p)
            return self
        if getattr(self, "grad_arena", None) is not None and len(self.grad_offsets) == len(ps) and self._arena_world == world:
            return self
        A = 64
        offs, off = [0] * len(ps), A                    # [0, 64): loss slot
        for i in reversed(range(len(ps))):
            offs[i] = off
            off += (ps[i].numel() + A - 1) // A * A
        q = A * max(1, world)
        total = (off + q - 1) // q * q
        buf = torch.zeros(total, dtype=torch.float32, device=ps[0].device)
        for p, o in zip(ps, offs):
            assert p.dtype == torch.float32 and p.is_contiguous()
            p.grad = buf[o:o + p.numel()].view_as(p)
        self.grad_arena, self.grad_offsets, self._arena_world = buf, offs, world
        return self

    def named_parameters(self):
        """(stable name, packed tensor) pairs: the diffusers name of the parameter an entry was packed from + the kind of the
        packed tensor, so optimizer state can be matched by NAME on resume (the registry's own order is the first-run order
        of the modules)"""
        names = {id(p): n for n, p in self.model.named_parameters()}
        out = []
        for e in self.gemms.values():
            base = names.get(id(e.weight), f"<unnamed weight {tuple(e.weight.shape)}>")
            for wp in e.weights[1:]:                  # a fused contraction: "...attn1.to_q.weight+to_k.weight+to_v.weight"
                base += "+" + ".".join(names.get(id(wp), "?").split(".")[-2:])
            out.append((base + "::packed", e.P))
            if e.Pb is not None:
                out.append((base + "::packed_bias", e.Pb))
        for a in self.affines.values():
            base = names.get(id(a.gamma_p), f"<unnamed affine {tuple(a.gamma_p.shape)}>")
            out += [(base + "::gamma", a.Pg), (base + "::beta", a.Pb)]
        return out

    def n_trainable(self) -> int:
        return sum(p.numel() for p in self.parameters())

    @torch.no_grad()
    def offload_masters_(self):
        """Move the diffusers-layout masters of the model to host memory (they are only the checkpoint interface while the
        packed state trains; ``export_`` writes into them wherever they live): -3.5 GB of HBM for an SD-2.1 expert."""
        for p in self.model.parameters():
            if p.device.type != "cpu":
                p.data = p.data.cpu()
        return self

    @torch.no_grad()
    def refresh_(self, shadows_done: bool = False):
        """shadows <- masters: one multi-tensor cast for the forward operands (skipped when the optimizer wrote them:
        PackedAdamW), ONE launch for all data-gradient operands (ops.PackDgradBatch: a table of the weights in device memory,
        rebuilt when an operand appears or moves)"""
        es = list(self.gemms.values())
        if not shadows_done:
            torch._foreach_copy_([e.pw.w for e in es], [e.P.detach() for e in es])
        pairs = [(e.pw, e.pwb) for e in es if e.pwb is not None]
        if not pairs:
            return
        key = tuple((pw.w.data_ptr(), pwb.w.data_ptr()) for pw, pwb in pairs)
        batch = self.__dict__.get("_dgrad_batch")
        if batch is None or batch.key != key:
            if torch.cuda.is_current_stream_capturing():
                # (the table is uploaded with a host->device copy: never while capturing; GraphedFineTunerStep's warm-up
                # iterations run first, so this only happens when a new operand shows up mid-capture)
                for e in es:
                    e.refresh_bwd_()
                return
            batch = self._dgrad_batch = ops.PackDgradBatch(pairs)
        batch.run()

    @torch.no_grad()
    def export_(self):
        """write the packed values back into the diffusers-named full-shape parameters (checkpoint interface)"""
        sync = getattr(self, "sync", None)
        if sync is not None:
            sync()                                # (a graphed step's optimizer tail runs on its own stream)
        for e in self.gemms.values():
            e.export_()
        for a in self.affines.values():
            a.export_()
        return self.model


class PackedAdamW:
    """torch.optim.AdamW's arithmetic over a PackedTrainer's state as ONE launch that also writes the bf16 operands
    (csrc/optim.hip).  The gradient tensors must exist and stay where they are (``.grad`` of every parameter as left by the
    captured backward of GraphedFineTunerStep: the graph re-writes the same addresses at every replay)."""

    def __init__(self, trainer: PackedTrainer, lr: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 group_of=None):
        """group_of: optional map parameter -> group index (e.g. the gradient bucket that completes it, ArenaGradReducer.bucket_of):
        the table is built per group and ``step_group(i)`` applies one group per launch, so the optimizer can follow a
        bucketed gradient exchange bucket by bucket; ``step()`` applies all groups."""
        import ctypes
        from . import _lib
        self.trainer, self.lr, self.betas, self.eps, self.weight_decay = trainer, lr, betas, eps, weight_decay
        self.group_of = group_of
        entries = []                                   # (parameter, bf16 shadow or None)
        for e in trainer.gemms.values():
            entries.append((e.P, e.pw.w))
            if e.Pb is not None:
                entries.append((e.Pb, None))
        for a in trainer.affines.values():
            entries += [(a.Pg, None), (a.Pb, None)]
        entries = [(p, sh) for p, sh in entries if p.grad is not None]      # (entries no backward reaches have nothing to apply)
        assert entries, "PackedAdamW: run one backward first (the gradient addresses go into the kernel's table)"
        dev = entries[0][0].device
        self.entries = entries
        self.params = [p for p, _ in entries]
        by_id = {id(p): n for n, p in trainer.named_parameters()} if hasattr(trainer, "named_parameters") else {}
        self.names = [by_id.get(id(p), f"#{i}") for i, p in enumerate(self.params)]
        self.m = [torch.zeros_like(p.data) for p in self.params]
        self.v = [torch.zeros_like(p.data) for p in self.params]
        self.step_t = torch.zeros((), dtype=torch.float32, device=dev)
        self.rebind_()

    def rebind_(self):
        """(re)build the kernel's table from the parameters' CURRENT gradient tensors (state is kept): once for a captured
        backward, whose replays re-write the same addresses; before every step of an eager loop, whose backward allocates"""
        from . import _lib
        lib = _lib.load()
        dev = self.params[0].device
        self.grads = [p.grad for p in self.params]      # keeps the addresses in the table alive
        items = (_lib.AdamWItem * len(self.entries))()
        starts = [0]
        for i, ((p, sh), m, v) in enumerate(zip(self.entries, self.m, self.v)):
            g = p.grad
            assert g is not None, f"PackedAdamW.rebind_: {self.names[i]} has no gradient (every tensor of the table needs one)"
            assert g.is_contiguous() and p.data.is_contiguous() and g.dtype == torch.float32 and p.dtype == torch.float32
            assert p.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0 and m.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0
            it = items[i]
            it.p, it.g, it.m, it.v, it.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
            if sh is not None:
                assert sh.dtype == torch.bfloat16 and sh.is_contiguous() and sh.numel() == p.numel() and sh.data_ptr() % 8 == 0
                it.shadow = sh.data_ptr()
            starts.append(starts[-1] + lib.aptp_adamw_blocks(p.numel()))
        self.items = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
        self.starts = torch.tensor(starts, dtype=torch.int32).to(dev)
        self.n, self.total = len(self.entries), starts[-1]
        # per-group tables (bucketed data-parallel exchange): the same descriptors, regrouped, with block prefixes of their own
        self.groups = None
        if self.group_of is not None:
            import ctypes
            isz = ctypes.sizeof(_lib.AdamWItem)
            raw = bytes(items)
            by = {}
            for i, p in enumerate(self.params):
                by.setdefault(int(self.group_of(p)), []).append(i)
            self.groups = {}
            for gi, idx in sorted(by.items()):
                st = [0]
                for i in idx:
                    st.append(st[-1] + (starts[i + 1] - starts[i]))
                blob = b"".join(raw[i * isz:(i + 1) * isz] for i in idx)
                self.groups[gi] = (torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev),
                                   torch.tensor(st, dtype=torch.int32).to(dev), len(idx), st[-1])
        return self

    def _launch(self, items, starts, n, total, gate, grad_scale):
        import ctypes
        from . import _lib
        lib = _lib.load()
        q = _lib.AdamWParams()
        q.items_dev, q.starts_dev, q.n_items, q.total_blocks = items.data_ptr(), starts.data_ptr(), n, total
        q.lr, q.beta1, q.beta2, q.eps, q.weight_decay = self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay
        q.step_dev = self.step_t.data_ptr()
        q.gate_dev = None if gate is None else gate.data_ptr()
        q.grad_scale = float(grad_scale)
        _lib.check(lib.aptp_adamw_many(ctypes.byref(q), ops._stream()), "aptp_adamw_many")

    def _check_grads(self):
        for p, g in zip(self.params, self.grads):
            assert p.grad is g, "PackedAdamW: a gradient tensor was replaced (its address is part of the kernel's table)"

    @torch.no_grad()
    def step(self, gate: torch.Tensor = None, grad_scale: float = 1.0):
        """one AdamW step over every tensor (one launch).  gate: optional fp32 device scalar (the step's loss): the whole step
        -- parameters, moments, operands and the step count -- is skipped unless it is finite (reference: the batch skip of
        pdm/training/trainer.py:921-929); grad_scale multiplies the gradients first (1 / world after a summed exchange)."""
        self._check_grads()
        self._launch(self.items, self.starts, self.n, self.total, gate, grad_scale)
        self.finish_step(gate)

    @torch.no_grad()
    def step_group(self, gi: int, gate: torch.Tensor = None, grad_scale: float = 1.0):
        """the tensors of group gi only (no step-count update: call finish_step() after the last group)"""
        if gi in self.groups:
            items, starts, n, total = self.groups[gi]
            self._launch(items, starts, n, total, gate, grad_scale)

    @torch.no_grad()
    def finish_step(self, gate: torch.Tensor = None):
        if gate is None:
            self.step_t += 1.0
        else:
            self.step_t += torch.isfinite(gate.reshape(())).to(torch.float32)
        self.trainer.refresh_(shadows_done=True)

    def state_dict(self):
        """step count and both moments, keyed by the stable names of PackedTrainer.named_parameters() (resume:
        load_state_dict on an optimizer built over the same expert, whatever order its modules first ran in)"""
        return {"step": self.step_t.clone(), "names": list(self.names),
                "exp_avg": [m.clone() for m in self.m], "exp_avg_sq": [v.clone() for v in self.v]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        assert len(sd["exp_avg"]) == len(sd["exp_avg_sq"]), "PackedAdamW: malformed state"
        names = sd.get("names")
        if names is None:                       # (states written before entries were named: positional)
            assert len(sd["exp_avg"]) == len(self.m), "PackedAdamW: different tensor list"
            order = list(range(len(self.m)))
        else:
            pos = {n: i for i, n in enumerate(names)}
            missing = [n for n in self.names if n not in pos]
            assert not missing and len(pos) == len(names) == len(self.names), \
                f"PackedAdamW: the saved state belongs to another expert / tensor list (missing {missing[:3]}, " \
                f"{len(names)} saved vs {len(self.names)} here)"
            order = [pos[n] for n in self.names]
        for i, j in enumerate(order):
            ma, va = sd["exp_avg"][j], sd["exp_avg_sq"][j]
            assert self.m[i].shape == ma.shape and self.v[i].shape == va.shape, \
                f"PackedAdamW: {self.names[i]}: saved moments {tuple(ma.shape)} / {tuple(va.shape)} vs {tuple(self.m[i].shape)}"
        self.step_t.copy_(sd["step"])
        for i, j in enumerate(order):
            self.m[i].copy_(sd["exp_avg"][j])    # in place: the moments' addresses are part of the kernel's table
            self.v[i].copy_(sd["exp_avg_sq"][j])
