"""Diagnostic: which part of the graphed fine-tune step breaks HIP graph capture.  Each case runs in its own process."""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = ["early_hooks_teacher_fwd_bwd_opt_refresh", "gft:simpleloss", "gft:"]


def cap(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()


def run(case):
    dev = torch.device("cuda:0")
    if case == "adamw":
        ps = [torch.nn.Parameter(torch.randn(1000, 64, device=dev)) for _ in range(40)]
        opt = torch.optim.AdamW(ps, lr=1e-3, fused=True, capturable=True)
        def fn():
            for p in ps: p.grad = torch.ones_like(p)
            opt.step()
        cap(fn)
    elif case == "foreach_copy":
        a = [torch.randn(300, 9, 64, device=dev) for _ in range(50)]; b = [torch.empty_like(t, dtype=torch.bfloat16) for t in a]
        cap(lambda: torch._foreach_copy_(b, a))
    elif case == "flipcopy":
        a = torch.randn(320, 9, 320, device=dev).bfloat16(); b = torch.zeros(320, 9, 320, device=dev, dtype=torch.bfloat16)
        cap(lambda: b[:320, :, :320].copy_(a[:, :, :320].flip(1).permute(2, 1, 0)))
    elif case == "wgrad_split":
        from diffusion_pruning_amd import ops
        x = torch.randn(2, 32, 32, 64, device=dev).bfloat16(); dy = torch.randn(2, 32, 32, 64, device=dev).bfloat16()
        cap(lambda: ops._wgrad_direct(x, dy, 3, 3, split_m=4))
    elif case.startswith("gft:"):
        from oracle import unet_oracle as O
        from diffusion_pruning_amd.train_step import GraphedFineTunerStep, synthetic_batch
        from diffusion_pruning_amd.unet import UNet2DConditionModelGated, UNet2DConditionModelPruned
        cfg = O.TINY
        kw = dict(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads, cross_attention_dim=cfg.cross_attention_dim)
        pm = UNet2DConditionModelPruned(**kw).init_synthetic(seed=0).to(dev)
        mask = O.random_mask(cfg, 0.6, 9, n_depth_off=1)
        pm.prune({k: [v.clone().to(dev) for v in vs] for k, vs in mask.items()})
        te = UNet2DConditionModelGated(**kw).init_synthetic(seed=0).to(dev)
        te.freeze()
        te.set_structure({k: [v.to(dev) for v in vs] for k, vs in O.ones_mask(cfg).items()})
        b = synthetic_batch(2, 16, dev, seed=4, cross_dim=cfg.cross_attention_dim)
        g = GraphedFineTunerStep(pm, te, lr=1e-4)
        g.capture(b, _diag=case[4:])
        for _ in range(3):
            out = g.train_step(None, b)
        torch.cuda.synchronize()
        print("loss", float(out["loss"]))
    else:
        from oracle import unet_oracle as O
        from diffusion_pruning_amd.packed_train import PackedTrainer
        from diffusion_pruning_amd.train_step import synthetic_batch
        from diffusion_pruning_amd.unet import UNet2DConditionModelPruned
        cfg = O.TINY
        pm = UNet2DConditionModelPruned(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                        cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0).to(dev)
        mask = O.random_mask(cfg, 0.6, 9, n_depth_off=1)
        pm.prune({k: [v.clone().to(dev) for v in vs] for k, vs in mask.items()})
        b = synthetic_batch(2, 16, dev, seed=4, cross_dim=cfg.cross_attention_dim)
        te = None
        if "early" in case:
            from diffusion_pruning_amd.train_step import FineTunerStep
            from diffusion_pruning_amd.unet import UNet2DConditionModelGated
            te = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                           cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0).to(dev)
            te.freeze()
            te.set_structure({k: [v.to(dev) for v in vs] for k, vs in O.ones_mask(cfg).items()})
            fts = FineTunerStep(pm, te)
        if "clones" in case:
            b = {k: v.clone() for k, v in b.items()}
        pk = PackedTrainer(pm).attach().materialize(b["noisy_latents"], b["timesteps"], b["encoder_hidden_states"])
        opt = torch.optim.AdamW(pk.parameters(), lr=1e-4, fused=True, capturable=True)
        keep = {}
        if "clones" in case:
            saved = [p.detach().clone() for p in pk.parameters()]
        if "hooks" in case and te is None:
            from diffusion_pruning_amd.train_step import FineTunerStep
            from diffusion_pruning_amd.unet import UNet2DConditionModelGated
            te = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                           cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0).to(dev)
            te.freeze()
            te.set_structure({k: [v.to(dev) for v in vs] for k, vs in O.ones_mask(cfg).items()})
            fts = FineTunerStep(pm, te)
        def fn():
            opt.zero_grad(set_to_none=True)
            if "teacher" in case:
                with torch.no_grad():
                    te(b["noisy_latents"], b["timesteps"], b["encoder_hidden_states"])
            out = pm(b["noisy_latents"], b["timesteps"], b["encoder_hidden_states"]).sample
            out.float().pow(2).mean().backward()
            if "opt" in case: opt.step()
            if "refresh" in case: pk.refresh_()
            if "_out_" in case: keep.update(total=out.detach())
        cap(fn)
    print("CASE", case, "ok", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, '-X', 'faulthandler', __file__, c], capture_output=True, text=True, timeout=170)
            tail = [t for t in (r.stdout + r.stderr).strip().splitlines() if 'amdgpu.ids' not in t and 'Extension modules' not in t and 'pluggy' not in t][-40:]
            print(f"== {c}: rc={r.returncode}\n" + "\n".join(t[:200] for t in tail), flush=True)
