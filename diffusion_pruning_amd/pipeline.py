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


@dataclass
class PipelineOutput:
    latents: torch.Tensor
    arch_indices: Optional[torch.Tensor]
    arch_vectors_quantized: Optional[torch.Tensor]
    resource_ratios: Optional[torch.Tensor] = None


class PruningDenoiseLoop:
    def __init__(self, unet, hyper_net=None, quantizer=None, scheduler: Optional[DDIMSchedulerLite] = None):
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

    def _one_step(self, latents, t, coef, ctx, guidance_scale, do_cfg):
        x = torch.cat([latents] * 2) if do_cfg else latents                       # pruning_pipelines.py:792
        noise = self.unet(x, t, ctx, return_dict=False)[0]                       # :796-802
        if do_cfg:
            uncond, text = noise.chunk(2)
            noise = uncond + guidance_scale * (text - uncond)                    # :805-807
        return self.scheduler.step_coef(noise, coef, latents)                    # :810-814

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
        if not use_graph:
            for i in range(num_inference_steps):
                latents = self._one_step(latents, ts[i].expand(B), self.scheduler.coef[i], ctx, guidance_scale, do_cfg)
        else:
            lat_buf = latents.clone()
            t_buf = ts[0].expand(B).clone()
            coef_buf = self.scheduler.coef[0].clone()
            # warm-up on a side stream (builds packed-weight plans), then capture one step
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._one_step(lat_buf, t_buf, coef_buf, ctx, guidance_scale, do_cfg)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._one_step(lat_buf, t_buf, coef_buf, ctx, guidance_scale, do_cfg)
            for i in range(num_inference_steps):
                t_buf.copy_(ts[i].expand(B))
                coef_buf.copy_(self.scheduler.coef[i])
                graph.replay()
                lat_buf.copy_(out)
            latents = lat_buf
        ratios = None
        if getattr(self.unet, "resource_info_dict", None) is not None:
            # pruning_pipelines.py:822-824
            ratios = self.unet.calc_macs()["cur_prunable_macs"] / self.unet.resource_info_dict["cur_prunable_macs"]
        return PipelineOutput(latents=latents, arch_indices=idx, arch_vectors_quantized=arch_q, resource_ratios=ratios)
