"""The pruning train step of APTP on synthetic batches (e): the counterpart of ``Pruner.step``
(pdm/training/trainer.py:1092-1254) plus the data-parallel gradient exchange that accelerate/DDP performs there
(:922).

What is reproduced: router (:1129-1138), cross-rank gather of text embeddings / normalised arch vectors with the local
slot re-inserted to keep autograd (:1147-1162), teacher forward with the all-ones structure under ``no_grad``
(:1185-1190), student forward with per-sample soft gates (:1192-1195), min-SNR weighted MSE with ``snr + 1`` for
v-prediction (:1197-1216), distillation + block-distillation on the 9 hooked block outputs (:496-511, 1218-1225),
resource / std / max losses from ``calc_macs`` (:1227-1238), and the loss weights of configs/pruning/sd-2-1_cc3m.yaml
(:86-111).  What is not: VAE / CLIP / MPNet encoders, datasets, logging, checkpointing (not on the U-Net path; the
batch arrives as latents, text states, MPNet embeddings and a target).

MI355X-first: one process per GPU; the trainable state is 1.26 M router parameters, so the per-step gradient
exchange is ONE flat fp32 all-reduce (~5 MB, latency-bound) over RCCL instead of DDP's bucket machinery, and the two
small all-gathers of the step are fused into one.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import dist_utils
from . import autograd as AG
from .graph_utils import concurrent_stream, new_graph, node_count
from . import ops
from .dist_utils import _rank, _span, _world
from .losses import ContrastiveLoss, ResourceLoss, compute_snr


@dataclass
class PruningLossConfig:
    """configs/pruning/sd-2-1_cc3m.yaml:86-111"""
    snr_gamma: Optional[float] = 5.0
    prediction_type: str = "v_prediction"
    resource_weight: float = 2.0
    resource_type: str = "log"
    pruning_target: float = 0.6
    contrastive_weight: float = 100.0
    arch_vector_temperature: float = 0.03
    prompt_embedding_temperature: float = 0.03
    distillation_weight: float = 0.2
    block_weight: float = 0.2
    std_weight: float = 0.1
    max_weight: float = 0.1


class NoiseSchedule:
    """scaled-linear DDPM betas of SD-2.1 (only ``alphas_cumprod`` is needed by compute_snr)"""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)


def gather_with_local_grad(*tensors: torch.Tensor) -> List[torch.Tensor]:
    """trainer.py:1147-1162: all-gather [B_loc, d_i] tensors across ranks under no_grad, re-insert the local block so
    gradients flow to the local rows only.  The tensors are concatenated along dim 1 so the step issues ONE collective."""
    world, rank = _world(), _rank()
    if world == 1:
        return list(tensors)
    widths = [t.shape[1] for t in tensors]
    with torch.no_grad():
        flat = torch.cat([t.detach().float() for t in tensors], dim=1).contiguous()
        gathered = torch.empty((world,) + tuple(flat.shape), dtype=flat.dtype, device=flat.device)
        with _span("all_gather(text, arch)"):
            dist.all_gather(list(gathered.unbind(0)), flat)      # (views of one buffer: no extra copy under RCCL)
    outs = []
    off = 0
    for t, w in zip(tensors, widths):
        blocks = [gathered[r, :, off:off + w].to(t.dtype) for r in range(world)]
        blocks[rank] = t
        outs.append(torch.cat(blocks, dim=0))
        off += w
    return outs


def allreduce_mean_grads(params) -> None:
    """One flat all-reduce (mean) of all router gradients: the DDP exchange of Pruner (1.26 M parameters,
    ~5 MB, latency-bound, so ONE collective instead of DDP's buckets).  For the 866 M parameters of an expert use
    BucketedGradReducer."""
    world = _world()
    if world == 1:
        return
    params = [p for p in params if p.requires_grad]
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
    with _span("all_reduce(router grads)"):
        dist.all_reduce(flat)
    flat /= world
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p).to(p.dtype)
        off += n


class BucketedGradReducer:
    """Data-parallel gradient exchange for the expert fine-tune (accelerate's DDP around
    trainer.py:1616): the 866 M parameter gradients are averaged in fixed-size buckets that are launched as soon as every
    gradient of a bucket has been accumulated, so the collectives overlap the rest of the backward.

    MI355X-first choices: buckets are sized for xGMI's per-link bound rings (default 64 MiB of payload: large enough to
    reach link bandwidth, small enough that the last bucket's exposed tail is a few ms), carried in bf16 (half the bytes
    on the 7 x 153 GB/s links; the fp32 master gradient receives the mean), and buckets follow REVERSE parameter
    registration order, which is the order the backward produces them in.  Works on any backend (gloo in the CPU tests)."""

    def __init__(self, params, bucket_bytes: int = 64 << 20, wire_dtype: torch.dtype = torch.bfloat16,
                 mode: str = "all_reduce", hooks: bool = True):
        """mode "all_reduce": one all-reduce per bucket.  mode "rs_ag": reduce-scatter + all-gather per bucket -- on the
        fully connected xGMI mesh of an MI355X node each of the two is ONE direct exchange between every pair of GPUs
        ((n-1)/n of the payload over 7 links at once), where a ring all-reduce is 2(n-1) dependent per-link steps (SURVEY
        5.8: 2.8 ms against 39.6 ms for the 1.7 GB of bf16 gradients); the all-gathers of all buckets are issued together
        after the backward.  hooks=False: no per-parameter hooks (a captured backward cannot run them): the caller
        exchanges everything with ``exchange_all()`` once the gradients are in place."""
        assert mode in ("all_reduce", "rs_ag"), mode
        self.params = [p for p in params if p.requires_grad]
        self.wire_dtype, self.mode = wire_dtype, mode
        self.buckets: List[List[torch.nn.Parameter]] = []
        esz = torch.empty((), dtype=wire_dtype).element_size()
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nb = p.numel() * esz
            if cur and cur_bytes + nb > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._shard: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._pending = [0] * len(self.buckets)
        self._work: List = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if hooks else []
        self.reset()

    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._work = []

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _on_grad(self, p):
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def _launch(self, i):
        world = _world()
        if world == 1:
            return
        ps = self.buckets[i]
        n = sum(p.numel() for p in ps)
        npad = (n + world - 1) // world * world          # (reduce-scatter needs equal shards; the padding is exchanged as zeros)
        flat = self._flat[i]
        if flat is None or flat.device != ps[0].device or flat.numel() != npad:
            flat = self._flat[i] = torch.zeros(npad, dtype=self.wire_dtype, device=ps[0].device)
            self._shard[i] = torch.empty(npad // world, dtype=self.wire_dtype, device=ps[0].device)
        off = 0
        for p in ps:
            k = p.numel()
            g = p.grad
            if g is None:
                flat[off:off + k].zero_()
            else:
                flat[off:off + k].copy_(g.reshape(-1))
            off += k
        self._pending[i] = 0
        if self.mode == "rs_ag":
            self._work.append((i, dist.reduce_scatter_tensor(self._shard[i], flat, async_op=True)))
        else:
            self._work.append((i, dist.all_reduce(flat, async_op=True)))

    def exchange_all(self):
        """launch every bucket now and finish (hooks=False: gradients written by a replayed HIP graph)"""
        self.reset()
        return self.finish()

    def finish(self):
        """Wait for every bucket, write the mean back into .grad (buckets whose parameters received no gradient this
        step are exchanged here so that all ranks issue the same collectives)."""
        world = _world()
        if world == 1:
            self.reset()
            return
        for i, left in enumerate(self._pending):
            if left > 0:
                self._launch(i)
        if self.mode == "rs_ag":
            # every shard is reduced once its reduce-scatter is done; the gathers of all buckets then go out back to back
            gathers = []
            for i, work in self._work:
                work.wait()
                gathers.append((i, dist.all_gather_into_tensor(self._flat[i], self._shard[i], async_op=True)))
            self._work = gathers
        inv = 1.0 / world
        for i, work in self._work:
            work.wait()
            flat, off = self._flat[i], 0
            for p in self.buckets[i]:
                k = p.numel()
                mean = flat[off:off + k].view_as(p)
                if p.grad is None:
                    p.grad = mean.to(p.dtype) * inv
                else:
                    p.grad.copy_(mean).mul_(inv)
                off += k
        self.reset()


class ArenaGradReducer:
    """Data-parallel gradient exchange of the graphed expert fine-tune (DDP inside accelerator.backward,
    pdm/training/trainer.py:1616) over ONE contiguous gradient arena (packed_train.PackedTrainer.ensure_grad_buffers(arena=True):
    every packed parameter's .grad is a view into it).  MI355X-first:
      * buckets are contiguous RANGES of the arena, so every collective runs in place on the memory the backward wrote and the
        optimizer reads: no staging buffer, no copy, no per-parameter kernel (the round-3 reducer issued ~3 tiny launches per packed
        tensor and step);
      * mode "rs_ag": reduce_scatter_tensor + all_gather_into_tensor per bucket, both in place (the rank's shard is a slice of the
        bucket) -- on the fully connected xGMI mesh each is ONE direct exchange between all pairs of GPUs over 7 links at once,
        where a ring all-reduce is 2(n-1) dependent per-link steps; mode "all_reduce": one collective per bucket;
      * the collectives are queued on a communication stream, bucket after bucket, and an event per bucket lets the consumer
        (the one-launch-per-bucket AdamW) start on bucket i while bucket i+1 is still on the links: what is exposed is one bucket,
        not the whole exchange;
      * the SUM stays in the arena: the 1 / world of the mean is AptpAdamWParams.grad_scale, folded into the optimizer's pass;
      * slot 0 of the arena carries the step's loss, so after the exchange it holds the sum over the ranks: one finite / non-finite
        verdict shared by all ranks (AptpAdamWParams.gate_dev) without a collective of its own.
    Issues its collectives whenever a process group is initialised -- also at world size 1 (bench.py under
    APTP_BENCH_FORCE_DIST: RCCL then runs every call of the schedule on one GPU)."""

    ALIGN = 64            # elements: bucket and shard boundaries are 256-byte aligned

    def __init__(self, arena: torch.Tensor, bucket_bytes: int = 256 << 20, mode: str = "rs_ag", group=None):
        assert mode in ("all_reduce", "rs_ag"), mode
        assert arena.dim() == 1 and arena.is_contiguous() and arena.dtype == torch.float32
        self.arena, self.mode, self.group = arena, mode, group
        self.world, self.rank = _world(), _rank()
        q = self.ALIGN * self.world
        assert arena.numel() % q == 0, "the arena is padded to a multiple of ALIGN * world elements (PackedTrainer does it)"
        per = max(q, (bucket_bytes // 4) // q * q)
        self.bounds: List[tuple] = []
        off = 0
        while off < arena.numel():
            end = min(arena.numel(), off + per)
            self.bounds.append((off, end))
            off = end
        self.buckets = [arena[a:b] for a, b in self.bounds]
        self.shards = [bk[(bk.numel() // self.world) * self.rank:(bk.numel() // self.world) * (self.rank + 1)] for bk in self.buckets]
        self.stream = torch.cuda.Stream(device=arena.device) if arena.is_cuda else None
        self.events = [torch.cuda.Event() for _ in self.buckets] if arena.is_cuda else None
        self.stats = {"collectives": 0, "tensor_ops": 0, "steps": 0}

    @property
    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized()

    def bucket_of(self, offset: int, numel: int) -> int:
        """bucket that completes the arena range [offset, offset + numel): the one holding its LAST element"""
        last = offset + numel - 1
        for i, (a, b) in enumerate(self.bounds):
            if a <= last < b:
                return i
        raise ValueError("range outside the arena")

    def exchange(self, on_bucket=None):
        """sum the arena over the ranks, bucket by bucket; on_bucket(i) is called once bucket i's sum is in place for the CURRENT
        stream (device side: the current stream waits for the bucket's event; nothing blocks the host on the GPU)"""
        self.stats["steps"] += 1
        if not self.active:
            for i in range(len(self.buckets)):
                if on_bucket is not None:
                    on_bucket(i)
            return
        cuda = self.stream is not None
        if cuda:
            main = torch.cuda.current_stream()
            self.stream.wait_stream(main)              # the gradients are complete on the launching stream
        ctx = torch.cuda.stream(self.stream) if cuda else _NullCtx()
        with ctx:
            for i, (bk, sh) in enumerate(zip(self.buckets, self.shards)):
                with _span("grad bucket %d" % i):
                    if self.mode == "rs_ag":
                        dist.reduce_scatter_tensor(sh, bk, group=self.group)
                        dist.all_gather_into_tensor(bk, sh, group=self.group)
                        self.stats["collectives"] += 2
                    else:
                        dist.all_reduce(bk, group=self.group)
                        self.stats["collectives"] += 1
                if cuda:
                    self.events[i].record(self.stream)
        for i in range(len(self.buckets)):
            if cuda:
                torch.cuda.current_stream().wait_event(self.events[i])
            if on_bucket is not None:
                on_bucket(i)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """mean((a - b)^2) in fp32 with the gradient flowing into ``a`` only (b: target / teacher side).  On the GPU the HIP
    reduction (autograd.MseFn: no fp32 copies, fixed order, safe inside replayed graphs); host tensors (the CPU suite's
    emulated runs) take torch's."""
    b = b.detach()
    if a.is_cuda:
        return AG.mse(a, b)
    return F.mse_loss(a.float(), b.float(), reduction="mean")


def _mse_per_sample(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[B] per-sample means (the min-SNR weighted diffusion loss, trainer.py:1203-1213)"""
    if a.is_cuda:
        return torch.stack([_mse(a[i], b[i]) for i in range(a.shape[0])])
    d = F.mse_loss(a.float(), b.detach().float(), reduction="none")
    return d.mean(dim=list(range(1, d.dim())))


class PrunerStep:
    def __init__(self, unet, hyper_net, quantizer, cfg: Optional[PruningLossConfig] = None,
                 schedule: Optional[NoiseSchedule] = None):
        self.unet, self.hyper_net, self.quantizer = unet, hyper_net, quantizer
        self.cfg = cfg or PruningLossConfig()
        self.schedule = schedule or NoiseSchedule()
        self.contrastive = ContrastiveLoss(self.cfg.arch_vector_temperature, self.cfg.prompt_embedding_temperature)
        self.resource = ResourceLoss(p=self.cfg.pruning_target, loss_type=self.cfg.resource_type)
        self.block_activations: Dict[str, torch.Tensor] = {}
        self._hooks = []
        self._cast_block_act_hooks()

    # trainer.py:496-511
    def _cast_block_act_hooks(self):
        acts = self.block_activations

        def mk(name, residuals_present):
            if residuals_present:
                return lambda m, i, o: acts.__setitem__(name, o[0])
            return lambda m, i, o: acts.__setitem__(name, o)
        for i, b in enumerate(self.unet.down_blocks):
            self._hooks.append(b.register_forward_hook(mk("d" + str(i), True)))
        self._hooks.append(self.unet.mid_block.register_forward_hook(mk("m", False)))
        for i, b in enumerate(self.unet.up_blocks):
            self._hooks.append(b.register_forward_hook(mk("u" + str(i), False)))

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @torch.no_grad()
    def count_macs(self, latent_size: int):
        """trainer.py:1256-1306: MAC constants with the all-ones structure, prunable-MAC template, actual target p."""
        ones = self.hyper_net.transform_structure_vector(
            torch.ones((1, self.quantizer.vq_embed_dim), device=next(self.hyper_net.parameters()).device))
        self.unet.set_structure(ones)
        self.unet.count_macs(latent_size)
        self.quantizer.set_prunable_macs_template([list(x) for x in self.unet.prunable_macs_list])
        info = self.unet.resource_info_dict
        p = self.cfg.pruning_target
        self.resource.p = float(1 - (1 - p) * info["total_macs"] / float(torch.as_tensor(info["cur_prunable_macs"]).flatten()[0]))

    def step(self, noisy_latents, timesteps, encoder_hidden_states, text_embeddings, target, pretrain: bool = False):
        cfg = self.cfg
        arch_vector = self.hyper_net(text_embeddings)                                           # :1129
        arch_vector_quantized, _ = self.quantizer(arch_vector)                                  # :1130
        arch_vector = self.quantizer.gumbel_sigmoid_trick(arch_vector)                          # :1132
        arch_vector = self._single_arch_repeat(arch_vector, text_embeddings.shape[0])           # :1134-1136
        arch_wdn = self.quantizer.width_depth_normalize(arch_vector)                            # :1138
        text_all, arch_all = gather_with_local_grad(text_embeddings, arch_wdn)                  # :1147-1162 (one collective)
        sep = self.hyper_net.transform_structure_vector(arch_vector if pretrain else arch_vector_quantized)   # :1165-1168
        contrastive_loss = self.contrastive(text_all, arch_all)                                 # :1170-1171

        with torch.no_grad():                                                                   # :1185-1190 teacher
            full = self.hyper_net.transform_structure_vector(torch.ones_like(arch_vector))
            self.unet.set_structure(full)
            full_pred = self.unet(noisy_latents, timesteps, encoder_hidden_states).sample.detach()
            teacher_acts = dict(self.block_activations)
        self.unet.set_structure(sep)                                                            # :1192-1195 student
        model_pred = self.unet(noisy_latents, timesteps, encoder_hidden_states).sample
        student_acts = dict(self.block_activations)

        if cfg.snr_gamma is None:                                                               # :1197-1216
            loss = _mse(model_pred, target)
        else:
            snr = compute_snr(self.schedule, timesteps)
            if cfg.prediction_type == "v_prediction":
                snr = snr + 1
            w = torch.stack([snr, cfg.snr_gamma * torch.ones_like(timesteps)], dim=1).min(dim=1)[0] / snr
            loss = (_mse_per_sample(model_pred, target) * w).mean()
        distillation_loss = _mse(model_pred, full_pred)  # :1218
        block_loss = torch.zeros((), device=model_pred.device)
        for k in student_acts:                                                                  # :1220-1225
            block_loss = block_loss + _mse(student_acts[k], teacher_acts[k])
        block_loss = block_loss / len(student_acts)

        macs = self.unet.calc_macs()                                                            # :1227-1238
        ratios = macs["cur_prunable_macs"] / self.unet.resource_info_dict["cur_prunable_macs"].squeeze()
        resource_loss = self.resource(ratios.mean())
        max_loss = 1.0 - torch.max(ratios)
        std_loss = -torch.std(ratios)

        diff_loss = loss.detach().clone()
        loss = loss + cfg.resource_weight * resource_loss + cfg.contrastive_weight * contrastive_loss \
            + cfg.distillation_weight * distillation_loss + cfg.block_weight * block_loss \
            + cfg.std_weight * std_loss + cfg.max_weight * max_loss                               # :1240-1249
        return {"loss": loss, "diff_loss": diff_loss, "distillation_loss": distillation_loss.detach(),
                "block_loss": block_loss.detach(), "contrastive_loss": contrastive_loss.detach(),
                "resource_loss": resource_loss.detach(), "resource_ratio": ratios.mean().detach(),
                "arch_vector_quantized": arch_vector_quantized.detach()}

    def _single_arch_repeat(self, arch_vector, batch: int):
        """trainer.py:1134-1136 (the single-architecture baseline): the ONE learned architecture vector serves the whole batch
        -- repeated after the Gumbel-sigmoid relaxation, so every sample sees the same noise draw -- and is remembered on the
        hyper-net as ``arch_gs``"""
        if getattr(self.hyper_net, "single_arch_param", False):
            arch_vector = arch_vector.repeat(batch, 1)
            self.hyper_net.arch_gs = arch_vector
        return arch_vector

    def trainable_parameters(self):
        return [p for p in list(self.hyper_net.parameters()) + list(self.quantizer.parameters()) if p.requires_grad]

    def train_step(self, optimizer, batch: dict, pretrain: bool = False):
        """forward + backward + fused gradient all-reduce + optimizer step (trainer.py:913-933)"""
        optimizer.zero_grad(set_to_none=True)
        out = self.step(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"],
                        batch["mpnet_embeddings"], batch["target"], pretrain=pretrain)
        out["loss"].backward()
        allreduce_mean_grads(self.trainable_parameters())
        optimizer.step()
        return out


class GraphedPrunerStep(PrunerStep):
    """PrunerStep with the two U-Net passes replayed from HIP graphs.

    The pruning step is host-bound when run eagerly (≈10^4 launches per step at SD-2.1 size: two U-Net forwards, one
    backward, autograd glue).  Everything that touches the U-Net has static shapes, so it is captured once:

    * ``g_teacher``: the all-ones (dense) forward under no_grad -> ``full_pred`` and the 9 block activations;
    * ``g_student``: forward with the architecture code read from STATIC gate buffers;
    * ``g_student_bwd``: the three U-Net loss terms (min-SNR diffusion MSE, output distillation, block distillation:
      trainer.py:1197-1225) and the backward down to the gradient of those terms w.r.t. every gate buffer.

    Each step the router runs eagerly (it draws host-side Gumbel noise, quirk Q6, and is tiny), its architecture code is
    copied into the static buffer the 84 gate views alias, the graphs are replayed, and the chain rule is closed eagerly:
    ``autograd.backward([router-only losses, arch code], [None, captured gradient w.r.t. the code])``.
    Losses and router gradients equal the eager step's (tests/test_train_step_gpu.py)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._cap = None
        self.stream_probe = []          # what graph_utils.concurrent_stream measured when it chose the teacher's / router's side streams
        self.overlap_router = os.environ.get("APTP_OVERLAP_ROUTER", "1") != "0"
        self._finish_hooks = []

    def remove_hooks(self):
        super().remove_hooks()
        for h in self._finish_hooks:
            h.remove()
        self._finish_hooks = []

    # ---- capture ------------------------------------------------------------------------------------------------
    def _snr_weights(self, timesteps):
        """min-SNR-gamma weights (trainer.py:1203-1213); the alpha-bar table lives on the device (_schedule_on), so the captured
        teacher graph computes them from the staged timesteps"""
        cfg = self.cfg
        if cfg.snr_gamma is None:
            return torch.ones(timesteps.shape[0], device=timesteps.device)
        snr = compute_snr(self.schedule, timesteps)
        if cfg.prediction_type == "v_prediction":
            snr = snr + 1
        return (torch.stack([snr, cfg.snr_gamma * torch.ones_like(timesteps)], dim=1).min(dim=1)[0] / snr).float()

    def _schedule_on(self, device):
        # keep the alpha-bar table on the device: a pageable host->device copy per step would make the host wait for the
        # previous step's graphs
        if self.schedule.alphas_cumprod.device != device:
            self.schedule.alphas_cumprod = self.schedule.alphas_cumprod.to(device)

    def _unet_losses(self, model_pred, student_acts, full_pred, teacher_acts, w, target):
        cfg = self.cfg
        if cfg.snr_gamma is None:
            loss = _mse(model_pred, target)
        else:
            loss = (_mse_per_sample(model_pred, target) * w).mean()
        distillation_loss = _mse(model_pred, full_pred)
        block_loss = torch.zeros((), device=model_pred.device)
        for k in student_acts:
            block_loss = block_loss + _mse(student_acts[k], teacher_acts[k])
        block_loss = block_loss / len(student_acts)
        return loss, distillation_loss, block_loss

    def capture(self, batch: dict, optimizer=None, pretrain: bool = False):
        """Capture the graphs for this batch geometry (call once; later batches must have the same shapes).
        optimizer: a CAPTURABLE optimizer over trainable_parameters() (torch.optim.AdamW(..., capturable=True)).  When given
        (and no process group is initialised: the router's collectives stay eager), the ROUTER is captured too -- hyper-net,
        quantiser incl. Sinkhorn, Gumbel relaxation, MAC losses as one graph ahead of the student's forward, and the chain rule
        into the router + the optimizer step as one graph behind the U-Net backward (trainer.py:1129-1138, :922-931) -- for this
        value of `pretrain`; its Gumbel noise keeps coming from the host generator (estimation_utils.NoiseTape)."""
        cfg = self.cfg
        dev = batch["noisy_latents"].device
        st = {k: batch[k].clone() for k in ("noisy_latents", "timesteps", "encoder_hidden_states", "target")}
        self._schedule_on(dev)
        st["snr_w"] = self._snr_weights(st["timesteps"])
        B = st["noisy_latents"].shape[0]
        ones = torch.ones((B, self.quantizer.vq_embed_dim), device=dev)
        full = self.hyper_net.transform_structure_vector(ones)
        # ONE static buffer holds the student's architecture code, SEGMENT-MAJOR: gate j's [B, w_j] block is contiguous, so
        # the 84 gate tensors the U-Net reads are contiguous autograd LEAVES aliasing it (a column slice of a [B, 1634]
        # matrix would make every kernel wrapper take a strided -> contiguous copy of its gate: 231 five-microsecond copies
        # per replay).  No autograd edge outside the captured region (a view node recorded on the legacy default stream would
        # make the captured backward hop streams).  A step installs a new code with one gather (`install_code`), and the
        # captured backward ends by concatenating the 84 gate gradients and gathering them back into [B, 1634] order.
        widths = list(self.hyper_net.width_list)
        n_depth = sum(self.hyper_net.depth_list)
        E = sum(widths) + n_depth
        assert E == self.quantizer.vq_embed_dim
        col = torch.arange(E)
        perm, s0 = [], 0
        for w in widths + [1] * n_depth:                          # position in the flat buffer -> index into arch.reshape(-1)
            perm.append((torch.arange(B)[:, None] * E + col[s0:s0 + w][None, :]).reshape(-1))
            s0 += w
        perm = torch.cat(perm)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel())
        perm, inv = perm.to(dev), inv.to(dev)
        ga = torch.ones(B * E, device=dev)
        gw, gd, s0 = [], [], 0
        for w in widths:
            gw.append(ga[s0:s0 + B * w].view(B, w).detach().requires_grad_(True))
            s0 += B * w
        for _ in range(n_depth):
            gd.append(ga[s0:s0 + B].detach().requires_grad_(True))
            s0 += B

        def install_code(arch):
            torch.index_select(arch.detach().reshape(-1), 0, perm, out=ga)
        from .macs import VectorizedMacs
        vmacs = VectorizedMacs(self.unet, device=dev)

        def teacher():
            # own split-K counters / scratch: this graph replays NEXT TO the student's forward (ops.scratch_domain)
            with torch.no_grad(), ops.scratch_domain("teacher"):
                # the min-SNR weights of this batch: a dozen tiny launches that only the loss terms read -- inside this graph they
                # run on the side stream next to the router instead of ahead of everything on the caller's stream (0.8 ms of a step)
                st["snr_w"].copy_(self._snr_weights(st["timesteps"]))
                pred = self.unet(st["noisy_latents"], st["timesteps"], st["encoder_hidden_states"]).sample.detach()
                return pred, dict(self.block_activations)

        def student_fwd():
            pred = self.unet(st["noisy_latents"], st["timesteps"], st["encoder_hidden_states"]).sample
            return pred, dict(self.block_activations)

        def student_bwd(pred, acts, full_pred, teacher_acts):
            loss, dist, blk = self._unet_losses(pred, acts, full_pred, teacher_acts, st["snr_w"], st["target"])
            total = loss + cfg.distillation_weight * dist + cfg.block_weight * blk
            grads = torch.autograd.grad(total, gw + gd, allow_unused=True)
            flat = torch.cat([(torch.zeros_like(t) if g is None else g).reshape(-1) for g, t in zip(grads, gw + gd)])
            grad = flat[inv].view(B, E)                        # back to the architecture vector's [B, 1634] order
            return loss.detach(), dist.detach(), blk.detach(), grad

        # warm-up on a side stream (allocator / plan caches), then capture: the module state at capture time decides what
        # is baked in, so the structure is installed right before each capture
        # ops.LAUNCH_LOG (bench.py: the contractions of ONE step, re-timed for the family roofline) is filled by this eager pass,
        # never by the captures: a log entry keeps its operands alive, and a capture that cannot recycle any activation spreads
        # the step over several times the memory -- its replays were measured ~7 % slower
        user_log = ops.LAUNCH_LOG
        log0 = None if user_log is None else len(user_log)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.unet.set_structure({"width": list(full["width"]), "depth": list(full["depth"])})
            fp, ta = teacher()
            self.unet.set_structure({"width": list(gw), "depth": list(gd)})
            student_bwd(*student_fwd(), fp, ta)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        launch_log = None if user_log is None else user_log[log0:]
        ops.LAUNCH_LOG = None

        # THREE graphs.  The student's forward does not read the teacher (only the loss terms do), so it is its own graph:
        # at replay it starts as soon as the router has produced the code, NEXT TO the teacher graph that has been running
        # on the side stream since the batch arrived; the loss + backward graph joins the two.  Graphs that replay
        # concurrently must not share a memory pool (a pool hands one capture's freed scratch to the next capture, which is
        # only safe when replays are serialised in capture order): the teacher has its own, the two student graphs share one.
        try:
            self.unet.set_structure({"width": list(full["width"]), "depth": list(full["depth"])})
            g_teacher = new_graph()
            with torch.cuda.graph(g_teacher):
                full_pred, teacher_acts = teacher()
            self.unet.set_structure({"width": list(gw), "depth": list(gd)})
            g_student = new_graph()
            with torch.cuda.graph(g_student):
                pred, acts = student_fwd()
            g_student_bwd = new_graph()
            with torch.cuda.graph(g_student_bwd, pool=g_student.pool()):
                loss, dist, blk, grad = student_bwd(pred, acts, full_pred, teacher_acts)
        finally:
            ops.LAUNCH_LOG = user_log
        # (everything a captured kernel reads must outlive the graphs: `inv` is an operand of the gather that ends g_student_bwd)
        self._cap = dict(st=st, ga=ga, install_code=install_code, perm=perm, inv=inv, full=full, pred=pred, acts=acts, teacher_acts=teacher_acts, gw=gw, gd=gd, g_teacher=g_teacher, g_student=g_student, g_student_bwd=g_student_bwd,
                         loss=loss, dist=dist, blk=blk, grad=grad, full_pred=full_pred, side=concurrent_stream(log=self.stream_probe), vmacs=vmacs,
                         launch_log=launch_log, router=None)
        if optimizer is not None and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self._capture_router(batch, optimizer, pretrain)       # (`dist` is the distillation loss in this scope)
        return self

    # ---- the router as two more graphs ----------------------------------------------------------------------------------
    def _router_forward(self, text_embeddings, pretrain: bool):
        """hyper-net -> quantiser -> Gumbel-sigmoid relaxation -> contrastive / MAC losses (step() up to the student's code)"""
        cfg, cap = self.cfg, self._cap
        arch_vector = self.hyper_net(text_embeddings)
        arch_vector_quantized, _ = self.quantizer(arch_vector)
        arch_vector = self.quantizer.gumbel_sigmoid_trick(arch_vector)
        arch_vector = self._single_arch_repeat(arch_vector, text_embeddings.shape[0])
        arch_wdn = self.quantizer.width_depth_normalize(arch_vector)
        text_all, arch_all = gather_with_local_grad(text_embeddings, arch_wdn)
        arch_used = arch_vector if pretrain else arch_vector_quantized                          # trainer.py:1165-1168
        contrastive_loss = self.contrastive(text_all, arch_all)
        macs = cap["vmacs"](arch_used)
        ratios = macs["cur_prunable_macs"] / self.unet.resource_info_dict["cur_prunable_macs"].squeeze()
        resource_loss = self.resource(ratios.mean())
        max_loss = 1.0 - torch.max(ratios)
        std_loss = -torch.std(ratios)
        router_loss = cfg.resource_weight * resource_loss + cfg.contrastive_weight * contrastive_loss \
            + cfg.std_weight * std_loss + cfg.max_weight * max_loss
        return dict(arch_used=arch_used, arch_vector_quantized=arch_vector_quantized, contrastive_loss=contrastive_loss,
                    resource_loss=resource_loss, ratios=ratios, router_loss=router_loss)

    def _capture_router(self, batch, optimizer, pretrain):
        from . import estimation_utils as EU
        cap = self._cap
        assert optimizer.defaults.get("capturable", False), "GraphedPrunerStep.capture(optimizer=): the optimizer must be capturable"
        params = self.trainable_parameters()
        text = batch["mpnet_embeddings"].clone()
        tape = EU.NoiseTape()
        saved_p = [p.detach().clone() for p in params]
        host_rng = torch.get_rng_state()

        def fwd():
            tape.begin()
            EU.NOISE_TAPE = tape
            try:
                r = self._router_forward(text, pretrain)
            finally:
                EU.NOISE_TAPE = None
            with torch.no_grad():
                cap["install_code"](r["arch_used"])
            return r

        def bwd(r):
            ts = [r["arch_used"]] if r["arch_used"].requires_grad else []
            torch.autograd.backward([r["router_loss"]] + ts, [None] + ([cap["grad"]] if ts else []))
            optimizer.step()

        # warm-up on a side stream: allocator, caches (gather indices, segment maps), the optimizer's state tensors; then undo it --
        # parameters, moments and step counts return to their values (in place: their addresses go into the graphs) and the host
        # generator to its state, so capturing costs the training run nothing
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                for p in params:
                    p.grad = None
                bwd(fwd())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, s0 in zip(params, saved_p):
                p.copy_(s0)
            for stt in optimizer.state.values():
                for v in stt.values():
                    if torch.is_tensor(v):
                        v.zero_()
        for p in params:
            p.grad = None
        g_fwd = new_graph()
        with torch.cuda.graph(g_fwd):
            r = fwd()
        g_bwd = new_graph()
        with torch.cuda.graph(g_bwd, pool=g_fwd.pool()):
            bwd(r)
        torch.set_rng_state(host_rng)
        cap["router"] = dict(g_fwd=g_fwd, g_bwd=g_bwd, out=r, text=text, tape=tape, optimizer=optimizer, pretrain=bool(pretrain),
                             stream=concurrent_stream(log=self.stream_probe), ev_entry=torch.cuda.Event(), ev_code=torch.cuda.Event(),
                             ev_bwd=torch.cuda.Event(), ev_tail=torch.cuda.Event(), pending=False)
        if not self._finish_hooks:
            # anything that reads the router's parameters outside the captured step first waits for the pending optimizer step
            for m in (self.hyper_net, self.quantizer):
                self._finish_hooks.append(m.register_forward_pre_hook(lambda *_a, **_k: self.finish()))
                self._finish_hooks.append(m.register_state_dict_pre_hook(lambda *_a, **_k: self.finish()))

    def graph_nodes(self):
        """nodes (kernel launches + torch's few memcpy / memset nodes) of the three captured graphs, when they were kept
        (graph_utils.KEEP_GRAPHS): {"teacher", "student_fwd", "student_bwd"}"""
        cap = self._cap
        if cap is None:
            return None
        n = {"teacher": node_count(cap["g_teacher"]), "student_fwd": node_count(cap["g_student"]),
             "student_bwd": node_count(cap["g_student_bwd"])}
        if cap.get("router"):
            n["router_fwd"] = node_count(cap["router"]["g_fwd"])
            n["router_bwd_and_optimizer"] = node_count(cap["router"]["g_bwd"])
        return n

    # ---- one step -------------------------------------------------------------------------------------------------------
    def step(self, noisy_latents, timesteps, encoder_hidden_states, text_embeddings, target, pretrain: bool = False):
        if self._cap is None:
            return super().step(noisy_latents, timesteps, encoder_hidden_states, text_embeddings, target, pretrain)
        cfg, cap = self.cfg, self._cap
        self._stage_batch_and_launch_teacher(noisy_latents, timesteps, encoder_hidden_states, target)
        arch_vector = self.hyper_net(text_embeddings)
        arch_vector_quantized, _ = self.quantizer(arch_vector)
        arch_vector = self.quantizer.gumbel_sigmoid_trick(arch_vector)
        arch_vector = self._single_arch_repeat(arch_vector, text_embeddings.shape[0])
        arch_wdn = self.quantizer.width_depth_normalize(arch_vector)
        text_all, arch_all = gather_with_local_grad(text_embeddings, arch_wdn)
        arch_used = arch_vector if pretrain else arch_vector_quantized                          # trainer.py:1165-1168
        contrastive_loss = self.contrastive(text_all, arch_all)

        # The teacher depends on the batch only: its graph replays on a side stream WHILE the router above (launch-bound
        # kernels of a few microseconds each, during which the GPU is mostly idle) is still being issued on this stream.
        # Order on the device: batch copies -> [teacher graph || router] -> student graph.
        # (The copies were queued first thing in this step, see _stage_batch_and_launch_teacher.)
        # MAC accounting is differentiable in the gates (calc_macs family): VectorizedMacs evaluates it on the router-connected
        # architecture vector directly (same numbers as unet.calc_macs() after set_structure, tests/test_macs_pinned.py) in
        # ~10 kernels instead of ~2,000, before the student replay is queued, so the host never waits on the graphs.
        macs = cap["vmacs"](arch_used)
        ratios = macs["cur_prunable_macs"] / self.unet.resource_info_dict["cur_prunable_macs"].squeeze()
        resource_loss = self.resource(ratios.mean())
        max_loss = 1.0 - torch.max(ratios)
        std_loss = -torch.std(ratios)

        with torch.no_grad():
            cap["install_code"](arch_used)
        cap["g_student"].replay()                                     # forward: needs the code only
        torch.cuda.current_stream().wait_stream(cap["side"])          # teacher outputs ready
        cap["g_student_bwd"].replay()                                 # losses against the teacher + backward to the code

        router_loss = cfg.resource_weight * resource_loss + cfg.contrastive_weight * contrastive_loss \
            + cfg.std_weight * std_loss + cfg.max_weight * max_loss
        unet_loss = cap["loss"] + cfg.distillation_weight * cap["dist"] + cfg.block_weight * cap["blk"]
        return {"loss": (router_loss.detach() + unet_loss).clone(), "diff_loss": cap["loss"].clone(),
                "distillation_loss": cap["dist"].clone(), "block_loss": cap["blk"].clone(),
                "contrastive_loss": contrastive_loss.detach(), "resource_loss": resource_loss.detach(),
                "resource_ratio": ratios.mean().detach(), "arch_vector_quantized": arch_vector_quantized.detach(),
                "_router_loss": router_loss, "_gate_tensors": [arch_used], "_gate_grads": [cap["grad"]]}

    def _stage_batch_and_launch_teacher(self, noisy_latents, timesteps, encoder_hidden_states, target):
        cap = self._cap
        with torch.no_grad():
            for k, src in (("noisy_latents", noisy_latents), ("timesteps", timesteps),
                           ("encoder_hidden_states", encoder_hidden_states), ("target", target)):
                cap["st"][k].copy_(src)
        side = cap["side"]                       # (the min-SNR weights are computed from the staged timesteps inside g_teacher)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            cap["g_teacher"].replay()

    @staticmethod
    def backward(out: dict):
        """Close the chain rule: d(total)/d(router) = d(router-only terms) + sum_g dL_unet/dgate_g * dgate_g/d(router)."""
        if "_router_loss" not in out:
            out["loss"].backward()
            return
        ts = [t for t in out["_gate_tensors"] if t.requires_grad]
        gs = [g for t, g in zip(out["_gate_tensors"], out["_gate_grads"]) if t.requires_grad]
        torch.autograd.backward([out["_router_loss"]] + ts, [None] + gs)

    def train_step(self, optimizer, batch: dict, pretrain: bool = False):
        cap = self._cap
        rt = cap.get("router") if cap is not None else None
        if rt is not None and rt["optimizer"] is optimizer and rt["pretrain"] == bool(pretrain) and self.hyper_net.training:
            return self._train_step_captured_router(batch)
        self.finish()
        optimizer.zero_grad(set_to_none=True)
        out = self.step(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"],
                        batch["mpnet_embeddings"], batch["target"], pretrain=pretrain)
        self.backward(out)
        allreduce_mean_grads(self.trainable_parameters())
        optimizer.step()
        return out

    def _train_step_captured_router(self, batch: dict):
        """the whole step from five graphs: [teacher || router] -> student forward -> losses + U-Net backward -> chain rule into the
        router + optimizer.  The host only stages the batch and refills the Gumbel uniforms from its generator.

        The two router graphs are chains of a few hundred tiny launches (0.9 + 1.3 ms during which the chip idles).  With
        overlap_router (default) they replay on a stream of their own: the NEXT step's batch staging and teacher forward do not
        queue behind this step's router backward + optimizer, they run next to it; the student's forward waits for the code
        (ev_code), the router backward for the U-Net backward (ev_bwd).  Reading router parameters on another stream afterwards
        needs finish() -- the router modules' forward / state_dict hooks and the eager train_step call it."""
        cfg, cap = self.cfg, self._cap
        rt = cap["router"]
        main = torch.cuda.current_stream()
        rs = rt["stream"] if self.overlap_router else main
        rt["ev_entry"].record(main)                           # the batch tensors were produced on the caller's stream
        self._stage_batch_and_launch_teacher(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"], batch["target"])
        with torch.cuda.stream(rs):
            rs.wait_event(rt["ev_entry"])
            with torch.no_grad():
                rt["text"].copy_(batch["mpnet_embeddings"])
            rt["tape"].refill()                               # host-RNG stream, consumed exactly as the eager calls consume it
            rt["g_fwd"].replay()                              # router -> architecture code installed for the student
            rt["ev_code"].record(rs)
        main.wait_event(rt["ev_code"])
        cap["g_student"].replay()
        main.wait_stream(cap["side"])
        cap["g_student_bwd"].replay()
        rt["ev_bwd"].record(main)
        with torch.cuda.stream(rs):
            rs.wait_event(rt["ev_bwd"])
            rt["g_bwd"].replay()                              # d(router losses + U-Net terms)/d(router), optimizer step
            rt["ev_tail"].record(rs)
        rt["pending"] = rs is not main
        r = rt["out"]
        # (the router graph's outputs were complete at ev_code; the next router forward, which overwrites them, waits for the next
        #  step's ev_entry, i.e. for these clones)
        unet_loss = cap["loss"] + cfg.distillation_weight * cap["dist"] + cfg.block_weight * cap["blk"]
        return {"loss": (r["router_loss"].detach() + unet_loss).clone(), "diff_loss": cap["loss"].clone(),
                "distillation_loss": cap["dist"].clone(), "block_loss": cap["blk"].clone(),
                "contrastive_loss": r["contrastive_loss"].detach().clone(), "resource_loss": r["resource_loss"].detach().clone(),
                "resource_ratio": r["ratios"].mean().detach(), "arch_vector_quantized": r["arch_vector_quantized"].detach().clone()}

    def finish(self):
        """make the current stream wait for the router backward + optimizer of the last captured step (they run on the router's own
        stream); called by the router modules' forward / state_dict hooks and by the eager paths of this class"""
        rt = self._cap.get("router") if self._cap is not None else None
        if rt is not None and rt.get("pending"):
            torch.cuda.current_stream().wait_event(rt["ev_tail"])
            rt["pending"] = False


def synthetic_batch(batch: int, latent: int, device, seed: int = 1234, cross_dim: int = 1024, text_dim: int = 768):
    """SURVEY §8d synthetic CC3M-shape batch: latents/target N(0,1), text states N(0,1), MPNet embeddings 0.05*N(0,1),
    random integer timesteps."""
    g = torch.Generator().manual_seed(seed)
    return {
        "noisy_latents": torch.randn(batch, 4, latent, latent, generator=g).to(device),
        "target": torch.randn(batch, 4, latent, latent, generator=g).to(device),
        "encoder_hidden_states": torch.randn(batch, 77, cross_dim, generator=g).to(device),
        "mpnet_embeddings": (0.05 * torch.randn(batch, text_dim, generator=g)).to(device),
        "timesteps": torch.randint(0, 1000, (batch,), generator=g).to(device),
    }


@dataclass
class FinetuneLossConfig:
    """configs/finetuning/sd-2-1_cc3m.yaml:86-95"""
    snr_gamma: Optional[float] = 5.0
    prediction_type: str = "v_prediction"
    diffusion_weight: float = 0.01
    block_weight: float = 0.5
    distillation_weight: float = 0.5


class FineTunerStep:
    """Expert fine-tuning step (pdm/training/trainer.py:1683-1765, config 5): teacher = dense ungated U-Net under
    no_grad, student = physically pruned expert (UNet2DConditionModelPruned) with every parameter trainable;
    loss = w_d * minSNR-MSE + w_b * block-MSE + w_k * distill-MSE.  Experts never communicate ("one expert per GPU" =
    independent processes, scripts/aptp/finetune.py:27-28); ``data_parallel=True`` covers the optional case of one
    expert on several GPUs with BucketedGradReducer."""

    def __init__(self, student, teacher, cfg: Optional[FinetuneLossConfig] = None, schedule: Optional[NoiseSchedule] = None,
                 data_parallel: bool = False, bucket_bytes: int = 64 << 20, reduce_mode: str = "all_reduce"):
        self.student, self.teacher = student, teacher
        # data_parallel: ONE expert trained on several GPUs: bf16 gradient buckets all-reduced while the
        # backward is still running; the default is the reference's "one expert per GPU", which never communicates
        self._dp, self._bucket_bytes, self._reduce_mode, self._reducer_packed = data_parallel, bucket_bytes, reduce_mode, False
        self.reducer = BucketedGradReducer([p for p in student.parameters() if p.requires_grad], bucket_bytes, mode=reduce_mode) \
            if data_parallel else None
        self.cfg = cfg or FinetuneLossConfig()
        self.schedule = schedule or NoiseSchedule()
        self.acts_s: Dict[str, torch.Tensor] = {}
        self.acts_t: Dict[str, torch.Tensor] = {}
        self._hooks = []
        for model, acts in ((student, self.acts_s), (teacher, self.acts_t)):
            def mk(name, residuals_present, acts=acts):
                if residuals_present:
                    return lambda m, i, o: acts.__setitem__(name, o[0])
                return lambda m, i, o: acts.__setitem__(name, o)
            for i, b in enumerate(model.down_blocks):
                self._hooks.append(b.register_forward_hook(mk("d" + str(i), True)))
            self._hooks.append(model.mid_block.register_forward_hook(mk("m", False)))
            for i, b in enumerate(model.up_blocks):
                self._hooks.append(b.register_forward_hook(mk("u" + str(i), False)))

    def step(self, noisy_latents, timesteps, encoder_hidden_states, target):
        cfg = self.cfg
        with torch.no_grad():
            full_pred = self.teacher(noisy_latents, timesteps, encoder_hidden_states).sample.detach()      # :1726-1727
        model_pred = self.student(noisy_latents, timesteps, encoder_hidden_states).sample                  # :1729
        if cfg.snr_gamma is None:
            loss = _mse(model_pred, target)
        else:
            snr = compute_snr(self.schedule, timesteps)
            if cfg.prediction_type == "v_prediction":
                snr = snr + 1
            w = torch.stack([snr, cfg.snr_gamma * torch.ones_like(timesteps)], dim=1).min(dim=1)[0] / snr
            loss = (_mse_per_sample(model_pred, target) * w).mean()
        diff_loss = loss.detach().clone()
        loss = loss * cfg.diffusion_weight
        block_loss = torch.zeros((), device=model_pred.device)
        if cfg.block_weight > 0:
            for k in self.acts_s:
                block_loss = block_loss + _mse(self.acts_s[k], self.acts_t[k])
            block_loss = block_loss / len(self.acts_s)
            loss = loss + cfg.block_weight * block_loss
        distillation_loss = _mse(model_pred, full_pred)
        loss = loss + cfg.distillation_weight * distillation_loss
        return {"loss": loss, "diff_loss": diff_loss, "distillation_loss": distillation_loss.detach(),
                "block_loss": block_loss.detach()}

    def _packed_reducer(self):
        """Under a PackedTrainer the gradients the optimizer consumes are those of the PACKED tensors (the diffusers-layout
        parameters receive none): the data-parallel exchange must run over them, or the ranks silently diverge."""
        pk = self.student.__dict__.get("_pk")
        if self._dp and pk is not None and not self._reducer_packed:
            assert pk.parameters(), "FineTunerStep(data_parallel=True): call PackedTrainer.materialize() before the first step"
            if self.reducer is not None:
                self.reducer.remove()
            self.reducer = BucketedGradReducer(pk.parameters(), self._bucket_bytes, mode=self._reduce_mode)
            self._reducer_packed = True

    def train_step(self, optimizer, batch: dict):
        self._packed_reducer()
        optimizer.zero_grad(set_to_none=True)
        out = self.step(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"], batch["target"])
        out["loss"].backward()
        if self.reducer is not None:
            self.reducer.finish()
        optimizer.step()
        pk = self.student.__dict__.get("_pk")
        if pk is not None:
            pk.refresh_()                     # packed masters (packed_train.py): one multi-tensor cast, no re-pack
        else:
            # the bf16 packs follow the fp32 masters through the parameters' version counters (unet._PlanCache): the next
            # forward re-packs what the optimizer changed; dropping the old packs now only returns their memory earlier
            self.student.invalidate_plans()
        return out


class GraphedFineTunerStep(FineTunerStep):
    """Expert fine-tuning with the trainable state in the kernels' layout (packed_train.PackedTrainer): dense teacher forward,
    pruned student forward, the three loss terms (trainer.py:1730-1763) and the backward replayed from THREE HIP graphs --
    teacher (side stream, own pool) next to the student's forward, then losses + backward -- followed by the batched weight
    gradients, the deferred folds, AdamW (trainer.py:1529-1540) and the refresh of the bf16 operands as a handful of launches
    (ops.WgradBatch, ops.FoldBatch, packed_train.PackedAdamW).  The ~10^4 launches of the eager step (which is host-bound)
    cost their device time only.  Same numbers as FineTunerStep on packed masters with the same optimizer, bit for bit
    (tests/test_finetune_gpu.py)."""

    def __init__(self, student, teacher, cfg: Optional[FinetuneLossConfig] = None, schedule: Optional[NoiseSchedule] = None,
                 lr: float = 1e-5, weight_decay: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-8,
                 data_parallel: bool = False, bucket_bytes: int = 256 << 20, reduce_mode: str = "rs_ag", nan_guard: bool = True):
        super().__init__(student, teacher, cfg, schedule)
        # data_parallel (one expert on several GPUs): the replayed graph and the batched weight-gradient launches
        # leave this rank's gradients in ONE contiguous arena; ArenaGradReducer sums it over the ranks in place, bucket by bucket
        # on a communication stream, and the optimizer follows bucket by bucket (the mean is its grad_scale)
        self._dp_graphed, self._bucket_bytes, self._reduce_mode = data_parallel, bucket_bytes, reduce_mode
        # nan_guard: a step whose loss is not finite leaves parameters, moments and the step count untouched (device-side:
        # AptpAdamWParams.gate_dev; reference: the batch skip of pdm/training/trainer.py:921-929)
        self.nan_guard = nan_guard
        self.defer_folds, self.direct_grads, self._folds = True, True, None
        self.defer_wgrads, self._wgrads = True, None
        self.overlap_teacher = os.environ.get("APTP_FT_OVERLAP_TEACHER", "1") != "0"   # teacher graph on a side stream next to the student's forward
        self.overlap_tail = os.environ.get("APTP_FT_OVERLAP_TAIL", "0") == "1"   # measured: 28.9 steps/s with, 29.9 without (HBM-bound tail next to the teacher: contention)
        from .packed_train import PackedTrainer
        self.trainer = PackedTrainer(student).attach()
        self.opt_kw = dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
        self.optimizer = None
        self._cap = None
        self.stream_probe = []          # what graph_utils.concurrent_stream measured when it chose the teacher's / router's side streams
        self.overlap_router = os.environ.get("APTP_OVERLAP_ROUTER", "1") != "0"
        self._finish_hooks = []

    def _losses(self, model_pred, full_pred, w, target):
        cfg = self.cfg
        if cfg.snr_gamma is None:
            loss = _mse(model_pred, target)
        else:
            loss = (_mse_per_sample(model_pred, target) * w).mean()
        diff = loss.detach()
        total = loss * cfg.diffusion_weight
        blk = torch.zeros((), device=model_pred.device)
        if cfg.block_weight > 0:
            for k in self.acts_s:
                blk = blk + _mse(self.acts_s[k], self.acts_t[k])
            blk = blk / len(self.acts_s)
            total = total + cfg.block_weight * blk
        dist_l = _mse(model_pred, full_pred)
        total = total + cfg.distillation_weight * dist_l
        return total, diff, dist_l.detach(), blk.detach()

    def _snr_weights(self, timesteps):
        cfg = self.cfg
        if cfg.snr_gamma is None:
            return torch.ones(timesteps.shape[0], device=timesteps.device)
        snr = compute_snr(self.schedule, timesteps)
        if cfg.prediction_type == "v_prediction":
            snr = snr + 1
        return (torch.stack([snr, cfg.snr_gamma * torch.ones_like(timesteps)], dim=1).min(dim=1)[0] / snr).float()

    def capture(self, batch: dict, warmup_iters: int = 2, offload_masters: bool = False):
        """Build the packed trainable state, the three HIP graphs (teacher forward | student forward | losses + backward) for this
        batch geometry, and the one-launch AdamW over the gradients the backward graph and the batched launches leave behind."""
        dev = batch["noisy_latents"].device
        st = {k: batch[k].clone() for k in ("noisy_latents", "timesteps", "encoder_hidden_states", "target")}
        if self.schedule.alphas_cumprod.device != dev:
            self.schedule.alphas_cumprod = self.schedule.alphas_cumprod.to(dev)
        st["snr_w"] = self._snr_weights(st["timesteps"])
        self.trainer.materialize(st["noisy_latents"], st["timesteps"], st["encoder_hidden_states"])
        # the block-output hooks kept the materialising forward's autograd graph (and through it every AccumulateGrad node)
        # alive: drop it before the warm-up builds the graphs the capture will re-use
        self.acts_s.clear()
        self.acts_t.clear()
        if offload_masters:
            self.trainer.offload_masters_()
            torch.cuda.empty_cache()
        params = self.trainer.parameters()
        out = {}
        # direct-gradient mode (ops.GRAD_DIRECT): persistent gradient buffers written by the kernels / the deferred folds; nothing
        # to reset between steps -- every parameter's live region is overwritten by each backward
        direct = self.direct_grads
        if direct:
            self.trainer.ensure_grad_buffers(arena=self._dp_graphed, world=_world())

        def teacher_fwd():
            # own split-K counters / scratch: this graph replays NEXT TO the previous step's optimizer tail (ops.scratch_domain)
            with torch.no_grad(), ops.scratch_domain("teacher"):
                st["snr_w"].copy_(self._snr_weights(st["timesteps"]))      # (min-SNR weights: read by the loss terms only)
                fp = self.teacher(st["noisy_latents"], st["timesteps"], st["encoder_hidden_states"]).sample.detach()
            return fp, dict(self.acts_t)

        def student_fwd():
            if not direct:
                for p in params:
                    p.grad = None
            pred = self.student(st["noisy_latents"], st["timesteps"], st["encoder_hidden_states"]).sample
            return pred, dict(self.acts_s)

        def student_bwd(pred, student_acts, full_pred, teacher_acts):
            self.acts_s.clear()
            self.acts_s.update(student_acts)
            self.acts_t.clear()
            self.acts_t.update(teacher_acts)
            total, diff, dist_l, blk = self._losses(pred, full_pred, st["snr_w"], st["target"])
            total.backward()
            out.update(total=total.detach(), diff=diff, dist=dist_l, blk=blk)

        def fwd_bwd():
            fp, ta = teacher_fwd()
            student_bwd(*student_fwd(), fp, ta)

        # warm-up (allocator, plan caches, autograd's stream anchors) runs forward + backward only: nothing to undo afterwards
        # (ops.LAUNCH_LOG -- bench.py: the contractions of ONE step -- is filled by the last eager iteration, never by the captures:
        #  see GraphedPrunerStep.capture)
        user_log, log0 = ops.LAUNCH_LOG, None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ops.GRAD_DIRECT = direct
        try:
            with torch.cuda.stream(side):
                for i in range(max(1, warmup_iters)):
                    if user_log is not None:
                        log0 = len(user_log)
                    fwd_bwd()
        finally:
            ops.GRAD_DIRECT = False
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        launch_log = None if user_log is None else user_log[log0:]
        ops.LAUNCH_LOG = None
        # TWO graphs.  The teacher's forward reads no trainable state, so step i+1's teacher pass does not have to wait for step
        # i's optimizer: it is its own graph (own memory pool: it replays NEXT TO the tail of the previous step, whose kernels
        # still read tensors of the student graph's pool), and train_step() issues the tail -- batched weight gradients, folds,
        # AdamW, operand refresh: 6.7 of 33.8 ms, HBM-bound for the most part -- on a second stream.
        g_teacher = new_graph()
        try:
            with torch.cuda.graph(g_teacher):
                full_pred, teacher_acts = teacher_fwd()
        except BaseException:
            ops.LAUNCH_LOG = user_log
            raise
        graph = new_graph()
        # The slab folds of the split weight gradients (and the chunk folds of the norm-affine gradients) are only RECORDED while
        # the backward is captured and run as ONE launch behind every replay (ops.FoldBatch: 357 launches of ~8 us otherwise):
        # nothing reads a parameter gradient before the optimizer.
        # Likewise the stride-1 weight gradients: recorded, then ONE launch per filter size behind the replay (ops.WgradBatch),
        # with pixel slices sized for the batch instead of for a chip-filling launch each (most weights: no slabs, no fold).
        ops.FOLD_DEFER = [] if (self.defer_folds and direct) else None
        ops.WGRAD_DEFER = [] if (self.defer_wgrads and self.defer_folds and direct) else None
        ops.GRAD_DIRECT = direct
        g_bwd = new_graph()
        try:
            # the student's forward does not read the teacher (only the loss terms do): its own graph, replayed NEXT TO the teacher
            # graph on the side stream; the loss + backward graph joins the two (same pool as the forward: replayed in capture order)
            with torch.cuda.graph(graph):
                pred, student_acts = student_fwd()
            with torch.cuda.graph(g_bwd, pool=graph.pool()):
                student_bwd(pred, student_acts, full_pred, teacher_acts)
            folds, wgrads = ops.FOLD_DEFER, ops.WGRAD_DEFER
        finally:
            ops.FOLD_DEFER = None
            ops.WGRAD_DEFER = None
            ops.GRAD_DIRECT = False
            ops.LAUNCH_LOG = user_log
        self._wgrads = ops.WgradBatch(wgrads) if wgrads else None
        self._folds = ops.FoldBatch(folds) if folds else None
        # The optimizer is ONE launch over every trainable tensor and writes the bf16 operands in the same pass
        # (packed_train.PackedAdamW, csrc/optim.hip); its table holds the addresses of the gradients the captured backward
        # left in `.grad`, which every replay re-writes in place.  It runs right behind the graph, followed by the one-launch
        # refresh of the data-gradient operands: three launches, no host work worth capturing.
        from .packed_train import PackedAdamW
        group_of = None
        if self._dp_graphed:
            assert direct, "the data-parallel graphed step exchanges the gradient arena of the direct-gradient mode"
            self.reducer = ArenaGradReducer(self.trainer.grad_arena, self._bucket_bytes, mode=self._reduce_mode)
            off = {id(p_): o for p_, o in zip(self.trainer.parameters(), self.trainer.grad_offsets)}
            group_of = lambda p_: self.reducer.bucket_of(off[id(p_)], p_.numel())          # noqa: E731
        self.optimizer = PackedAdamW(self.trainer, lr=self.opt_kw["lr"], betas=self.opt_kw["betas"], eps=self.opt_kw["eps"],
                                     weight_decay=self.opt_kw["weight_decay"], group_of=group_of)
        self._cap = dict(st=st, graph=graph, g_bwd=g_bwd, g_teacher=g_teacher, full_pred=full_pred, teacher_acts=teacher_acts,
                         pred=pred, student_acts=student_acts, side=concurrent_stream(log=self.stream_probe),
                         tail_stream=torch.cuda.Stream(), ev_bwd=torch.cuda.Event(), ev_tail=None, launch_log=launch_log, **out)
        self.trainer.sync = self.finish          # (export_ / state_dict read the parameters: after the pending optimizer tail)
        return self

    def finish(self):
        """make the current stream wait for the optimizer tail of the last train_step (it runs on its own stream so that the next
        step's teacher forward overlaps it); call before reading parameters on the current stream -- PackedTrainer.export_ does"""
        cap = self._cap
        if cap is not None and cap.get("ev_tail") is not None:
            torch.cuda.current_stream().wait_event(cap["ev_tail"])

    def graph_nodes(self):
        """nodes of the captured teacher + student forward / backward graph, when it was kept (graph_utils.KEEP_GRAPHS)"""
        if self._cap is None:
            return None
        n = {"teacher": node_count(self._cap["g_teacher"]), "student_fwd": node_count(self._cap["graph"]),
             "student_bwd": node_count(self._cap["g_bwd"])}
        if self._wgrads is not None:
            n["batched_wgrad_launches"] = self._wgrads.launches()
        if self._folds is not None:
            n["deferred_folds_launch"] = 1
        return n

    def train_step(self, optimizer=None, batch: Optional[dict] = None):
        """one replayed step on `batch` (same shapes as the captured one); `optimizer` is ignored: the fused AdamW built at
        capture time is part of the graph"""
        cap = self._cap
        assert cap is not None, "call capture(batch) first"
        with torch.no_grad():
            for k in ("noisy_latents", "timesteps", "encoder_hidden_states", "target"):
                cap["st"][k].copy_(batch[k])
        main = torch.cuda.current_stream()       # (the min-SNR weights are computed from the staged timesteps inside g_teacher)
        side = cap["side"] if self.overlap_teacher else main
        side.wait_stream(main)                   # (the batch copies above)
        with torch.cuda.stream(side):
            cap["g_teacher"].replay()            # reads no trainable state; own pool, own scratch domain: next to the student's forward
        if cap["ev_tail"] is not None:
            main.wait_event(cap["ev_tail"])      # parameters and operands of the previous step are in place
        cap["graph"].replay()                    # student forward
        main.wait_stream(side)                   # teacher outputs ready
        cap["g_bwd"].replay()                    # losses against the teacher + backward
        cap["ev_bwd"].record(main)
        tail = cap["tail_stream"] if self.overlap_tail else main
        with torch.cuda.stream(tail):
            tail.wait_event(cap["ev_bwd"])
            if self._wgrads is not None:
                self._wgrads.run()               # every stride-1 weight gradient of the backward: one launch per filter size
            if self._folds is not None:
                self._folds.run()                # every deferred slab / chunk fold of the backward: one launch
            if self._dp_graphed and self.reducer is not None:
                # sum over the ranks in place, bucket by bucket on the communication stream; AdamW follows bucket by bucket with
                # grad_scale = 1 / world; arena slot 0 = sum of the ranks' losses = their common finite / non-finite verdict
                arena = self.trainer.grad_arena
                arena[0:1].copy_(cap["total"].reshape(1))
                gate = arena[0:1] if self.nan_guard else None
                scale = 1.0 / self.reducer.world
                self.reducer.exchange(lambda i: self.optimizer.step_group(i, gate, scale))
                self.optimizer.finish_step(gate)
            else:
                self.optimizer.step(cap["total"] if self.nan_guard else None)
            ev = cap["ev_tail"] or torch.cuda.Event()
            ev.record(tail)
            cap["ev_tail"] = ev
        # (clones: the graph's static outputs are overwritten by the next replay, and callers accumulate losses over steps)
        return {"loss": cap["total"].clone(), "diff_loss": cap["diff"].clone(), "distillation_loss": cap["dist"].clone(),
                "block_loss": cap["blk"].clone()}
