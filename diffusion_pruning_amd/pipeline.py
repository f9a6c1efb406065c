"""Denoise-loop glue of StableDiffusionPruningPipeline (pdm/pipelines/pruning_pipelines.py:746-759, 787-824), SURVEY §8
row a21 / f.2: route the prompt batch once -> ``unet.set_structure`` -> per step: CFG batch doubling -> U-Net ->
``uncond + s*(text - uncond)`` -> scheduler step.  VAE / CLIP / safety checker are not on the U-Net path and are not
reproduced (synthetic latents and text states, BASELINE.json).

MI355X-first differences:
  * the cross-attention K/V projections depend only on the text states, so they are computed ONCE per prompt batch
    (``UNet2DConditionModelGated.precompute_context``) and reused by every step;
  * one step (U-Net + CFG combine + DDIM update, all on the device, timestep and scheduler coefficients read from
    device tensors) is captured into a HIP graph and replayed ``num_inference_steps`` times;
  * a hard, batch-shared architecture code (one expert per call, as in generate_fid_images.py) takes the
    compacted-weight fast path of the U-Net; per-prompt codes fall back to fused per-sample gates.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


class DDIMSchedulerLite:
    """Minimal DDIM (eta = 0) for SD-2.1's schedule: scaled-linear betas 0.00085..0.012 over 1000 steps, "leading"
    timestep spacing with steps_offset 1, v-prediction or epsilon.  Plain tensor math (device-agnostic), restated
    from the published DDIM update; the reference uses diffusers' DDIM/PNDM schedulers (pruning_pipelines.py:805-814)."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 prediction_type: str = "v_prediction", steps_offset: int = 1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (torch.arange(0, num_inference_steps) * ratio).round().flip(0).long() + self.steps_offset
        self.num_inference_steps = num_inference_steps
        self.timesteps = ts.to(device) if device is not None else ts
        prev = ts - ratio
        a_t = self.alphas_cumprod[ts]
        a_prev = torch.where(prev >= 0, self.alphas_cumprod[prev.clamp(min=0)], self.final_alpha_cumprod)
        # per-step coefficient table [steps, 4]: sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)
        self.coef = torch.stack([a_t.sqrt(), (1 - a_t).sqrt(), a_prev.sqrt(), (1 - a_prev).sqrt()], dim=1)
        if device is not None:
            self.coef = self.coef.to(device)
        return self.timesteps

    def step_coef(self, model_output: torch.Tensor, coef: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
        """x_{t-1} from the model output with coefficients coef = [sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)]"""
        sa, sb, sap, sbp = coef[0], coef[1], coef[2], coef[3]
        if self.prediction_type == "v_prediction":
            x0 = sa * sample - sb * model_output
            eps = sa * model_output + sb * sample
        else:
            eps = model_output
            x0 = (sample - sb * eps) / sa
        return sap * x0 + sbp * eps

    # ---- per-step state interface shared with PNDMSchedulerLite (everything a step reads lives in static device tensors, so
    # one captured step serves the whole loop) -------------------------------------------------------------------------------
    def n_model_calls(self) -> int:
        return self.num_inference_steps

    def make_state(self, latents: torch.Tensor) -> dict:
        return {"coef": self.coef[0].clone()}

    def load_step(self, state: dict, i: int):
        state["coef"].copy_(self.coef[i])

    def step(self, model_output: torch.Tensor, sample: torch.Tensor, state: dict) -> torch.Tensor:
        return self.step_coef(model_output, state["coef"], sample)


class PNDMSchedulerLite:
    """PNDM / PLMS as StableDiffusionPruningPipeline runs it (configs/img_generation/sd-2-1_cc3m.yaml:50; scheduler call
    pruning_pipelines.py:810-814): pseudo linear multi-step with ``skip_prk_steps=True``, scaled-linear betas, "leading"
    spacing with steps_offset 1, epsilon or v-prediction.  N inference steps are N + 1 U-Net calls: the second call repeats
    the second timestep and redoes the first transfer with the average of the two outputs (the PLMS warm start), then
    2-, 3- and 4-term Adams-Bashforth combinations of the stored outputs feed the transfer
        x_prev = sqrt(a_prev / a_t) x - (a_prev - a_t) eps / (a_t sqrt(1 - a_prev) + sqrt(a_t (1 - a_t) a_prev)).
    PARITY PIN: diffusers==0.23.1 is absent, so this is restated from the published algorithm (Liu et al., "Pseudo Numerical
    Methods for Diffusion Models on Manifolds", 2022, eq. 9 and 12; diffusers' ``scheduling_pndm.py`` step_plms /
    _get_prev_sample) -- unpinned against a live run; tests pin it against DDIMSchedulerLite (identical transfer for a
    constant model output) and against a plain-Python restatement with explicit history lists.
    Graph-friendly form: the history is a 5-slot device ring, and which slot is written, the combination weights, the
    transfer coefficients and the two warm-start switches are per-step TABLES copied into static buffers, so one captured
    step serves all N + 1 calls."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 prediction_type: str = "v_prediction", steps_offset: int = 1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        N = num_inference_steps
        ratio = self.num_train_timesteps // N
        base = (torch.arange(0, N) * ratio).round().long() + self.steps_offset          # ascending
        plms = torch.cat([base[:-1], base[-2:-1], base[-1:]]).flip(0)                    # [t_N-1, t_N-2, t_N-2, t_N-3, ..., t_0]
        self.num_inference_steps = N
        self.timesteps = plms.to(device) if device is not None else plms
        # per-call tables: slot written (0..3 history ring, 4 = scratch), weights over the 5 slots, (a_t, a_prev) of the
        # transfer, use_saved (call 1 restarts from the sample saved at call 0), save (call 0 saves its sample)
        slots, W, coef, use_saved, save = [], [], [], [], []
        hist = []                                      # ring slots of the stored outputs, newest last
        nxt = 0
        for c in range(N + 1):
            t = int(plms[c])
            w = [0.0] * 5
            if c == 1:
                slot, t_from, t_to = 4, t + ratio, t
                w[hist[-1]], w[4] = 0.5, 0.5
            else:
                slot, t_from, t_to = nxt, t, t - ratio
                nxt = (nxt + 1) % 4
                hist = (hist + [slot])[-4:]
                if len(hist) == 1:
                    w[hist[-1]] = 1.0
                elif len(hist) == 2:
                    w[hist[-1]], w[hist[-2]] = 1.5, -0.5
                elif len(hist) == 3:
                    w[hist[-1]], w[hist[-2]], w[hist[-3]] = 23.0 / 12, -16.0 / 12, 5.0 / 12
                else:
                    w[hist[-1]], w[hist[-2]], w[hist[-3]], w[hist[-4]] = 55.0 / 24, -59.0 / 24, 37.0 / 24, -9.0 / 24
            a_t = float(self.alphas_cumprod[t_from])
            a_p = float(self.alphas_cumprod[t_to]) if t_to >= 0 else float(self.final_alpha_cumprod)
            slots.append(slot); W.append(w); coef.append([a_t, a_p])
            use_saved.append(1.0 if c == 1 else 0.0); save.append(1.0 if c == 0 else 0.0)
        self.tab = {"slot": torch.tensor(slots, dtype=torch.long), "w": torch.tensor(W, dtype=torch.float32),
                    "coef": torch.tensor(coef, dtype=torch.float32),
                    "flags": torch.tensor(list(zip(use_saved, save)), dtype=torch.float32)}
        if device is not None:
            self.tab = {k: v.to(device) for k, v in self.tab.items()}
        return self.timesteps

    def n_model_calls(self) -> int:
        return self.num_inference_steps + 1

    def make_state(self, latents: torch.Tensor) -> dict:
        z = torch.zeros_like(latents, dtype=torch.float32)
        return {"slot": self.tab["slot"][0:1].clone(), "w": self.tab["w"][0].clone(), "coef": self.tab["coef"][0].clone(),
                "flags": self.tab["flags"][0].clone(), "E": torch.zeros((5,) + tuple(latents.shape), dtype=torch.float32, device=latents.device),
                "saved": z}

    def load_step(self, state: dict, i: int):
        state["slot"].copy_(self.tab["slot"][i:i + 1])
        state["w"].copy_(self.tab["w"][i])
        state["coef"].copy_(self.tab["coef"][i])
        state["flags"].copy_(self.tab["flags"][i])

    def step(self, model_output: torch.Tensor, sample: torch.Tensor, state: dict) -> torch.Tensor:
        E, w, flags = state["E"], state["w"], state["flags"]
        x = sample.float()
        E.index_copy_(0, state["slot"], model_output.float()[None])
        state["saved"].add_(flags[1] * (x - state["saved"]))                 # call 0: remember the sample
        base = x + flags[0] * (state["saved"] - x)                           # call 1: restart from it
        comb = (w.view(5, *([1] * x.dim())) * E).sum(dim=0)
        a_t, a_p = state["coef"][0], state["coef"][1]
        if self.prediction_type == "v_prediction":
            comb = a_t.sqrt() * comb + (1 - a_t).sqrt() * base
        denom = a_t * (1 - a_p).sqrt() + (a_t * (1 - a_t) * a_p).sqrt()
        out = (a_p / a_t).sqrt() * base - (a_p - a_t) * comb / denom
        return out.to(sample.dtype)


@dataclass
class PipelineOutput:
    latents: torch.Tensor
    arch_indices: Optional[torch.Tensor]
    arch_vectors_quantized: Optional[torch.Tensor]
    resource_ratios: Optional[torch.Tensor] = None


class PruningDenoiseLoop:
    def __init__(self, unet, hyper_net=None, quantizer=None, scheduler=None):
        self.unet, self.hyper_net, self.quantizer = unet, hyper_net, quantizer
        self.scheduler = scheduler or DDIMSchedulerLite()
        self._graph = None
        self._graph_key = None

    @torch.no_grad()
    def route(self, hyper_net_input: torch.Tensor):
        """pruning_pipelines.py:746-759: hyper_net -> quantizer (eval: cosine assignment, hard code) -> split -> set"""
        self.hyper_net.eval()
        self.quantizer.eval()
        arch = self.hyper_net(hyper_net_input)
        arch_q, (_, _, idx) = self.quantizer(arch)
        sep = self.hyper_net.transform_structure_vector(arch_q)
        self.unet.set_structure(sep)
        return arch_q, idx

    def _one_step(self, latents, t, state, ctx, guidance_scale, do_cfg):
        x = torch.cat([latents] * 2) if do_cfg else latents                       # pruning_pipelines.py:792
        noise = self.unet(x, t, ctx, return_dict=False)[0]                       # :796-802
        if do_cfg:
            uncond, text = noise.chunk(2)
            noise = uncond + guidance_scale * (text - uncond)                    # :805-807
        return self.scheduler.step(noise, latents, state)                        # :810-814

    @torch.no_grad()
    def __call__(self, prompt_embeds: torch.Tensor, latents: torch.Tensor, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, hyper_net_input: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, use_graph: bool = True) -> PipelineOutput:
        """prompt_embeds [B,77,X] (+ negative_prompt_embeds for CFG, concatenated as [uncond, cond] like the
        reference, :765); latents [B,4,h,w] ~ N(0,1) on the device."""
        dev = latents.device
        arch_q = idx = None
        if self.hyper_net is not None and hyper_net_input is not None:
            arch_q, idx = self.route(hyper_net_input.to(dev))
        do_cfg = guidance_scale > 1.0 and negative_prompt_embeds is not None
        ehs = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds
        ctx = self.unet.precompute_context(ehs.to(dev))                           # cross-attn K/V once per prompt batch
        ts = self.scheduler.set_timesteps(num_inference_steps, device=dev)
        latents = latents * self.scheduler.init_noise_sigma
        B = latents.shape[0] * (2 if do_cfg else 1)
        n_calls = self.scheduler.n_model_calls()
        if not use_graph:
            state = self.scheduler.make_state(latents)
            for i in range(n_calls):
                self.scheduler.load_step(state, i)
                latents = self._one_step(latents, ts[i].expand(B), state, ctx, guidance_scale, do_cfg)
        else:
            # one captured step per (scheduler, shapes, CFG, guidance, installed architecture): later calls with the same key
            # (the FID-generation loop: many prompt batches through one expert) only refresh the static buffers
            key = (type(self.scheduler).__name__, self.scheduler.prediction_type, tuple(latents.shape), latents.dtype, do_cfg,
                   float(guidance_scale), str(dev), ctx.key, getattr(self.unet, "_structure_epoch", None))
            if ctx.key is None or self._graph_key != key:
                self._graph = self._capture(latents, ts, ctx, B, guidance_scale, do_cfg)
                self._graph_key = key if ctx.key is not None else None
            g = self._graph
            g["lat"].copy_(latents)
            g["ctx"].ehs.copy_(ctx.ehs)
            for k_, v_ in ctx.kv.items():
                g["ctx"].kv[k_].copy_(v_)
            for name, t_ in self.scheduler.make_state(latents).items():
                g["state"][name].copy_(t_)
            for i in range(n_calls):
                g["t"].copy_(ts[i].expand(B))
                self.scheduler.load_step(g["state"], i)
                g["graph"].replay()
                g["lat"].copy_(g["out"])
            latents = g["lat"].clone()
        ratios = None
        if getattr(self.unet, "resource_info_dict", None) is not None:
            # pruning_pipelines.py:822-824
            ratios = self.unet.calc_macs()["cur_prunable_macs"] / self.unet.resource_info_dict["cur_prunable_macs"]
        return PipelineOutput(latents=latents, arch_indices=idx, arch_vectors_quantized=arch_q, resource_ratios=ratios)

    def _capture(self, latents, ts, ctx, B, guidance_scale, do_cfg):
        lat_buf = latents.clone()
        t_buf = ts[0].expand(B).clone()
        state = self.scheduler.make_state(latents)
        self.scheduler.load_step(state, 0)
        # warm-up on a side stream (builds packed-weight plans), then capture one step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._one_step(lat_buf, t_buf, state, ctx, guidance_scale, do_cfg)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self._one_step(lat_buf, t_buf, state, ctx, guidance_scale, do_cfg)
        return {"graph": graph, "lat": lat_buf, "t": t_buf, "state": state, "out": out, "ctx": ctx}
