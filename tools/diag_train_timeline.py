"""Device timeline of one pruning step with the router captured (GraphedPrunerStep.capture(optimizer=)): events on the launch
stream after every graph, the teacher's span on its side stream, and each router graph replayed alone.
usage: python3 tools/diag_train_timeline.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from diffusion_pruning_amd.hypernet import HyperStructure
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
from diffusion_pruning_amd.train_step import GraphedPrunerStep, synthetic_batch
from diffusion_pruning_amd.unet import UNet2DConditionModelGated

dev = torch.device("cuda:0")
unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
unet.freeze()
st = unet.get_structure()
torch.manual_seed(0)
hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(dev)
qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3, depth_order=[-1, -2, 0, 1, -3, -4, 2, 3, -5, -6, 4, 5, -7, 6],
                              resource_aware_normalization=False, optimal_transport=True, fused_sinkhorn_allreduce=True).to(dev)
hn.train(); qz.train()
step = GraphedPrunerStep(unet, hn, qz)
step.count_macs(64)
opt = torch.optim.AdamW(step.trainable_parameters(), lr=2e-4, capturable=True)
batch = synthetic_batch(4, 64, dev)
step.capture(batch, optimizer=opt)
cap, rt = step._cap, step._cap["router"]
print("side-stream probe:", step.stream_probe, flush=True)


def alone(g, n=20):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# (the router graphs are not replayed out of order: g_bwd takes an optimizer step, and a router driven off its trajectory can feed
#  torch's binary_cross_entropy a value outside [0, 1] -- a device-side assert, i.e. a GPU exception)
print(f"alone: teacher {alone(cap['g_teacher']):.2f}  student fwd {alone(cap['g_student']):.2f}  student bwd {alone(cap['g_student_bwd']):.2f} ms", flush=True)
names = ["staged", "router fwd done", "student fwd done", "teacher joined", "student bwd done", "router bwd + opt done"]
N = 20
acc = [0.0] * len(names)
t_teacher = 0.0
for it in range(N + 3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    te0, te1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    ev[0].record()
    with torch.no_grad():
        for k in ("noisy_latents", "timesteps", "encoder_hidden_states", "target"):
            cap["st"][k].copy_(batch[k])
        cap["st"]["snr_w"].copy_(step._snr_weights(batch["timesteps"]))
        rt["text"].copy_(batch["mpnet_embeddings"])
    side = cap["side"]
    side.wait_stream(main)
    with torch.cuda.stream(side):
        te0.record(side)
        cap["g_teacher"].replay()
        te1.record(side)
    rt["tape"].refill()
    ev[1].record()
    rt["g_fwd"].replay(); ev[2].record()
    cap["g_student"].replay(); ev[3].record()
    main.wait_stream(side); ev[4].record()
    cap["g_student_bwd"].replay(); ev[5].record()
    rt["g_bwd"].replay(); ev[6].record()
    torch.cuda.synchronize()
    if it >= 3:
        for i in range(len(names)):
            acc[i] += ev[0].elapsed_time(ev[i + 1])
        t_teacher += te0.elapsed_time(te1)
print("timeline (ms since the step's first event, mean of %d steps): " % N + "  ".join(f"{n} {a / N:.2f}" for n, a in zip(names, acc))
      + f"  | teacher graph on its stream {t_teacher / N:.2f} ms", flush=True)
import time
t0 = time.perf_counter()
for _ in range(N):
    step.train_step(opt, batch)
torch.cuda.synchronize()
print(f"train_step: {(time.perf_counter() - t0) / N * 1e3:.2f} ms per step", flush=True)
t0 = time.perf_counter()
for _ in range(N):
    rt["tape"].refill()
t1 = time.perf_counter()
print(f"host: NoiseTape.refill {1e3 * (t1 - t0) / N:.3f} ms", flush=True)
# host cost of the graph launches themselves (hipGraphLaunch returns after enqueueing): is the step host-bound?
import time as _t
for name, g in (("teacher", cap["g_teacher"]), ("student fwd", cap["g_student"]), ("student bwd", cap["g_student_bwd"])):
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = _t.perf_counter()
        g.replay()
        ts.append((_t.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    print(f"host time of one replay() call, device idle: {name}: {min(ts):.2f} ms (nodes: {step.graph_nodes().get(name.replace(' ', '_'))})", flush=True)
torch.cuda.synchronize()
t0 = _t.perf_counter()
for _ in range(10):
    step.train_step(opt, batch)
t1 = _t.perf_counter()
torch.cuda.synchronize()
t2 = _t.perf_counter()
print(f"10 train_steps: host issue {1e2 * (t1 - t0):.2f} ms per step, wall {1e2 * (t2 - t0):.2f} ms per step", flush=True)
