"""Where the graphed pruning train step spends its time: each phase alone (synchronised), then the whole step.
usage: python tools/diag_pruner_phases.py [--batch 4] [--latent 64]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from diffusion_pruning_amd.hypernet import HyperStructure
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
from diffusion_pruning_amd.train_step import GraphedPrunerStep, synthetic_batch
from diffusion_pruning_amd.unet import UNet2DConditionModelGated


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3      # host-issue ms, wall ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--pageable-noise", action="store_true", help="move the Gumbel noise with a pageable copy (A/B)")
    a = ap.parse_args()
    if a.pageable_noise:
        from diffusion_pruning_amd import estimation_utils as EU
        EU._PinnedRing.send = lambda self, g, device: g.to(device)
    dev = torch.device("cuda:0")
    unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
    unet.freeze()
    st = unet.get_structure()
    torch.manual_seed(0)
    hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(dev)
    qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3,
                                  depth_order=[-1, -2, 0, 1, -3, -4, 2, 3, -5, -6, 4, 5, -7, 6],
                                  resource_aware_normalization=False, optimal_transport=True, fused_sinkhorn_allreduce=True).to(dev)
    hn.train(); qz.train()
    step = GraphedPrunerStep(unet, hn, qz)
    step.count_macs(a.latent)
    opt = torch.optim.AdamW(step.trainable_parameters(), lr=2e-4)
    batch = synthetic_batch(a.batch, a.latent, dev)
    step.capture(batch)
    cap = step._cap
    te = batch["mpnet_embeddings"]

    def router_fwd():
        av = hn(te)
        avq, _ = qz(av)
        av2 = qz.gumbel_sigmoid_trick(av)
        wdn = qz.width_depth_normalize(av2)
        c = step.contrastive(te, wdn)
        m = cap["vmacs"](avq)
        r = m["cur_prunable_macs"] / unet.resource_info_dict["cur_prunable_macs"].squeeze()
        return c + step.resource(r.mean()) - torch.std(r) + (1 - torch.max(r)), avq

    def router_all():
        opt.zero_grad(set_to_none=True)
        loss, avq = router_fwd()
        torch.autograd.backward([loss, avq], [None, cap["grad"]])
        opt.step()

    av0 = hn(te).detach()
    pieces = [("  hyper-net forward", lambda: hn(te)),
              ("  quantizer forward", lambda: qz(av0)),
              ("  gumbel_sigmoid_trick", lambda: qz.gumbel_sigmoid_trick(av0)),
              ("  width_depth_normalize", lambda: qz.width_depth_normalize(av0)),
              ("  contrastive loss", lambda: step.contrastive(te, av0)),
              ("  vectorized MACs", lambda: cap["vmacs"](av0))]
    rows = pieces + [("teacher graph", lambda: cap["g_teacher"].replay()),
            ("student forward graph", lambda: cap["g_student"].replay()),
            ("student losses + backward graph", lambda: cap["g_student_bwd"].replay()),
            ("router forward (eager)", router_fwd),
            ("router fwd + bwd + AdamW (eager)", router_all),
            ("whole train_step", lambda: step.train_step(opt, batch))]
    for name, fn in rows:
        h, w = timed(fn)
        print(f"{name:40s} host-issue {h:7.3f} ms   wall {w:7.3f} ms")


if __name__ == "__main__":
    main()
