"""Host-issue time vs wall time of the graphed fine-tune step's pieces (hipGraphLaunch of a ~2,900-node graph is not free:
16 ms of host time per replay on MI355X / ROCm 7.2, against 35 ms of device time).  usage: python tools/diag_ft_host.py"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import torch
from bench_finetune import expert_mask
from diffusion_pruning_amd.train_step import GraphedFineTunerStep, synthetic_batch
from diffusion_pruning_amd.unet import UNet2DConditionModelGated, UNet2DConditionModelPruned
dev = torch.device("cuda:0")
teacher = UNet2DConditionModelGated().init_synthetic(seed=0)
student = UNet2DConditionModelPruned(); student.load_state_dict(teacher.state_dict())
teacher.to(dev).freeze(); st = teacher.get_structure()
teacher.set_structure({"width": [torch.ones(1, w, device=dev) for sub in st["width"] for w in sub], "depth": [torch.ones(1, device=dev) for sub in st["depth"] for d in sub if d == 1]})
student.to(dev); student.prune(expert_mask(st, 0, dev))
batch = synthetic_batch(4, 64, dev)
step = GraphedFineTunerStep(student, teacher, lr=1e-5); step.capture(batch, offload_masters=True)
for _ in range(3): step.train_step(None, batch)
torch.cuda.synchronize()
t0 = time.perf_counter(); 
for _ in range(10): step.train_step(None, batch)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"graph.replay: host-issue {(t1-t0)/10*1e3:.2f} ms, wall {(t2-t0)/10*1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(10): step.optimizer.step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"optimizer.step (+refresh): host {(t1-t0)/10*1e3:.2f} ms, wall {(t2-t0)/10*1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(10): step.train_step(None, batch)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"train_step: host {(t1-t0)/10*1e3:.2f} ms, wall {(t2-t0)/10*1e3:.2f} ms")
