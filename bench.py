"""Headline benchmark: masked SD-2.1 U-Net denoise steps per second on MI355X (BASELINE.json configs[1]), plus the
data-parallel pruning train step (configs[2]/[3]) behind ``--config train``.

``python bench.py --gpus N --steps K --warmup W``
    N == 1 (or WORLD_SIZE already set by torchrun): this process is a rank.
    N  > 1 and WORLD_SIZE unset: this process only LAUNCHES -- it starts
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...``
    as a child BEFORE anything touches the GPU, relays rank 0's JSON line and exits with the child's code.
    Rank 0 reports ``n_gpus`` as the result of a device all-reduce of ones, so the record proves RCCL saw N ranks.

--config infer (default)
    A "step" is one UNet2DConditionModelGated.forward on a [4,4,64,64] latent batch (512x512 images, bs=4), fixed 50 %
    channel mask, synthetic latents / text states / seeded random-init weights, inputs resident in HBM.
    The forward is captured once into a HIP graph and replayed; K replays are timed between barrier +
    torch.cuda.synchronize().  Multi-GPU = independent replicas (inference shards by prompt, no data-path
    collective): MAX time over ranks, value = all ranks' steps / that time.
--config train
    A "step" is one APTP pruning train step (Pruner.step, trainer.py:1092-1254, on a synthetic CC3M-shape batch of bs=4
    per GPU): router, dense teacher forward, soft-masked student forward + backward to the 84 gate gradients (both
    replayed from HIP graphs), losses, chain rule into the router, ONE fused all-gather (text embeddings + normalised
    architecture vectors, trainer.py:1152-1154), the distributed Sinkhorn assignment (quantizer.py:284-291, fused into
    one all-gather) and ONE flat all-reduce of the router gradients (the DDP exchange of trainer.py:922), AdamW.
    Reports steps/s summed over ranks, per-GPU steps/s and the collectives' share of a step.

--config finetune
    BASELINE configs[4]: the expert fine-tune step (FineTuner.step, trainer.py:1683-1765 -- dense teacher forward, pruned
    student forward + backward incl. weight gradients, AdamW) on eight distinct seeded architecture codes (keep ratio
    0.40 .. 0.75, 0-4 depth gates off), ONE EXPERT PER GPU: rank r trains expert (r + --expert-offset) % 8 and the ranks
    never communicate (scripts/aptp/finetune.py:27-28,40); N = 1 trains expert 3.  The whole step replays from a HIP graph
    (train_step.GraphedFineTunerStep).  value = steps/s summed over the ranks (each rank's own rate is listed).

The default line (--config infer, one GPU) also carries:
  sustained       >= 3 s of back-to-back replays of the same graph: steps/s, the slowest / median 100-step window and the
                  shader clock (sysfs pp_dpm_sclk, sampled by a host thread while the replays run) beside it.
  extra_configs   BASELINE configs[2] (train) and configs[4] (finetune, expert 3) measured after the headline in this same
                  process: steps/s, ms/step, graph nodes (launches) per step, the conv_gemm family's MFMA roofline
                  fraction inside that step, peak memory.

Prints ONE JSON line (see the task contract) with extra objects:
  roofline        MFMA roofline of the dominant kernel family (conv_gemm*: implicit-GEMM conv/linear): achieved =
                  algorithmic FLOPs of all its launches in one forward / their measured device time (HIP events on the
                  launch stream around a HIP graph that replays exactly those launches); peak = 2500 TFLOP/s dense bf16
                  (MI355X_MICROARCH.md); peak_measured = what a plain 8192^3 bf16 GEMM reaches on THIS box (hipBLASLt via
                  torch.matmul, and this repo's own kernel) and a 1 GiB device copy; per_tile = the same fraction for each
                  kernel instantiation.
  roofline_groupnorm  HBM roofline of the GroupNorm(+SiLU) family: algorithmic bytes (x read once + y written once) / time.
  roofline_attention  MFMA roofline of the attention launches of one forward (4 B h Lq Lk 64 FLOP each), self- and cross-attention apart.
  gpu_vendor_baseline  the oracle's op sequence on this GPU in bf16 through torch-ROCm's own libraries (MIOpen / hipBLASLt / SDPA),
                  gated semantics, HIP-graph replay (tools/bench_vendor.py, child process): a stated yardstick, not the target.
  cpu_baseline    the oracle (PyTorch CPU restatement, fp32) timed on this host's cores on a bounded sample: median of 5
                  steps of the headline workload (bs=4, masked) and of BASELINE configs[0] (bs=1, dense).
"""
import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=("infer", "train", "finetune"), default="infer")
    ap.add_argument("--expert-offset", type=int, default=None, help="(finetune) rank r trains expert (r + offset) %% 8; default 3 at N = 1, else 0")
    ap.add_argument("--data-parallel", action="store_true", help="(finetune) all ranks train ONE expert (expert --expert-offset), gradients "
                    "summed in place over the gradient arena (train_step.ArenaGradReducer), mean folded into AdamW")
    ap.add_argument("--eager-router", action="store_true", help="(train) keep the router eager between the U-Net graphs (A/B against the captured router)")
    ap.add_argument("--no-vendor-baseline", action="store_true", help="(infer) skip the torch-ROCm vendor-library yardstick (gpu_vendor_baseline)")
    ap.add_argument("--no-extra-configs", action="store_true", help="(infer) skip the train / finetune measurements that follow the headline")
    ap.add_argument("--sustain-seconds", type=float, default=3.0, help="(infer) length of the sustained-replay leg; 0 = off")
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch")
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--dense", action="store_true", help="(infer) mask == 1 instead of the fixed 50 %% mask")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline legs (timing only)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo with --dryrun-cpu)")
    ap.add_argument("--dryrun-cpu", action="store_true",
                    help="TEST ONLY: run the launch / rendezvous / timing / reporting logic on CPU with the emulated ops of "
                         "tests/ on a tiny model (gloo); the printed line is marked as not a measurement")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 50 if args.config == "infer" else 10
    if args.warmup is None:
        args.warmup = 5 if args.config == "infer" else 3
    if args.expert_offset is None:
        args.expert_offset = 3 if args.gpus == 1 else 0
    return args


# ----------------------------------------------------------------------------------------------------------------------
# launcher: no torch.cuda call, no HIP call, nothing but a child process
# ----------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv) -> int:
    """Start N ranks under torchrun as a CHILD process (never exec: a process that has touched the GPU must not be
    replaced, and this one has not touched it at all), relay rank 0's JSON line, return the child's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    got = False
    for line in proc.stdout:
        ok = False
        if line.startswith("{"):
            try:
                ok = "metric" in json.loads(line)
            except ValueError:
                ok = False
        if ok and not got:
            got = True
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and not got:
        sys.stderr.write("bench.py: the ranks exited without printing a result line\n")
        return 1
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# rank
# ----------------------------------------------------------------------------------------------------------------------
class Rank:
    def __init__(self, args):
        import torch
        self.torch = torch
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: launch with "
                             f"--nproc-per-node {args.gpus} (or let bench.py launch the ranks itself)")
        self.cpu = args.dryrun_cpu
        if self.cpu:
            self.dev = torch.device("cpu")
        else:
            self.dev = torch.device("cuda", self.local_rank)
            torch.cuda.set_device(self.dev)
        self.n_seen = 1
        # APTP_BENCH_FORCE_DIST=1: bring the process group up even for one rank (exercises RCCL init / all-reduce /
        # barrier on a one-GPU box)
        self.dist = self.world > 1 or os.environ.get("APTP_BENCH_FORCE_DIST") == "1"
        if self.dist:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1:                         # forced single-rank group without a launcher: fill in the rendezvous
                os.environ.setdefault("MASTER_PORT", "29533")
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            backend = args.backend or ("gloo" if self.cpu else "nccl")
            kw = {} if self.cpu else {"device_id": self.dev}
            dist.init_process_group(backend=backend, **kw)
            ones = torch.ones(1, device=self.dev)
            dist.all_reduce(ones)                       # the record's n_gpus is what the collective library saw
            self.n_seen = int(ones.item())
            assert self.n_seen == self.world, (self.n_seen, self.world)

    def sync(self):
        if not self.cpu:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.dist:
            self.torch.distributed.barrier()

    def max_over_ranks(self, seconds: float) -> float:
        if not self.dist:
            return seconds
        te = self.torch.tensor([seconds], device=self.dev, dtype=self.torch.float64)
        self.torch.distributed.all_reduce(te, op=self.torch.distributed.ReduceOp.MAX)
        return float(te.item())

    def timed(self, run, steps, warmup):
        """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
        for _ in range(warmup):
            run()
        self.sync()
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        self.sync()
        self.last_local_elapsed = time.perf_counter() - t0        # this rank's own K steps (before the closing barrier)
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def finish(self):
        if self.dist:
            self.torch.distributed.destroy_process_group()


def fixed_half_mask(structure, device):
    """every 32-wide gate keeps even entries, head gates keep the first floor(h/2) heads, all depth gates on"""
    import torch
    width = []
    for sub in structure["width"]:
        for w in sub:
            g = torch.zeros(1, w, device=device)
            if w == 32:
                g[:, 0::2] = 1.0
            else:
                g[:, :max(1, w // 2)] = 1.0
            width.append(g)
    depth = [torch.ones(1, device=device) for sub in structure["depth"] for d in sub if d == 1]
    return {"width": width, "depth": depth}


def ones_mask(structure, device):
    import torch
    return {"width": [torch.ones(1, w, device=device) for sub in structure["width"] for w in sub],
            "depth": [torch.ones(1, device=device) for sub in structure["depth"] for d in sub if d == 1]}


def algorithmic_flops(masked: bool, batch: int, latent: int) -> float:
    """matmul-only FLOPs of one forward: 223.67 GMAC masked / 402.13 GMAC dense per sample @64x64"""
    assert latent == 64
    return 2.0 * (223.66507008e9 if masked else 402.12668416e9) * batch


def _dryrun_install():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests import bench_dryrun
    return bench_dryrun


def run_infer(R: Rank):
    torch, args, dev = R.torch, R.args, R.dev
    from diffusion_pruning_amd import ops
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated

    if R.cpu:
        model_kw = _dryrun_install().install_infer()
    else:
        model_kw = {}
    model = UNet2DConditionModelGated(**model_kw).init_synthetic(seed=0).to(dev)
    structure = model.get_structure()
    model.set_structure(ones_mask(structure, dev) if args.dense else fixed_half_mask(structure, dev))

    g = torch.Generator(device="cpu").manual_seed(1234 + R.rank)
    B, L = args.batch, args.latent
    xdim = model.config["cross_attention_dim"]
    sample = torch.randn(B, 4, L, L, generator=g).to(dev)
    ehs = torch.randn(B, 77, xdim, generator=g).to(dev)
    t = torch.full((B,), 500, dtype=torch.int64, device=dev)

    def step():
        return model(sample, t, ehs, return_dict=False)[0]

    roofline = roofline_gn = roofline_attn = cpu_baseline = vendor_baseline = None
    with torch.no_grad():
        out = step()                      # builds the packed-weight plans
        R.sync()
        assert torch.isfinite(out).all()
        gout = None
        if args.no_graph or R.cpu:
            run = step
        else:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gout = step()
            run = graph.replay
        elapsed = R.timed(run, args.steps, args.warmup)
        if gout is not None:
            # the timed replays computed the same thing as the eager forward above (deterministic kernels, same inputs)
            d = float((gout.float() - out.float()).abs().max())
            assert torch.isfinite(gout).all() and d <= 1e-2 * float(out.float().abs().max()), f"graph replay differs from eager: {d}"
        ms_per_step = elapsed / args.steps * 1e3
        value = R.world * args.steps / elapsed
        sustained = None
        if R.rank == 0 and not R.cpu and not args.no_extras and gout is not None and R.world == 1 and args.sustain_seconds > 0:
            sustained = measure_sustained(run, args.sustain_seconds, value, R.local_rank)
        if R.rank == 0 and not R.cpu and not args.no_extras:
            # (at N > 1 the other ranks wait for these legs in the final barrier: only the short ones run there)
            roofline = measure_roofline(ops, step, dev, per_tile=R.world == 1)
            if L == 64:
                flops = algorithmic_flops(not args.dense, B, L)
                roofline["whole_step_tflops"] = round(flops / (ms_per_step * 1e-3) / 1e12, 1)
            if R.world == 1:
                roofline["peak_measured"] = measure_peaks(ops, dev)
            roofline_gn = measure_gn_roofline(ops, step, dev)
            roofline_attn = measure_attn_roofline(ops, step, dev)
            if R.world == 1 and not args.no_cpu_baseline:
                cpu_baseline = measure_cpu_baseline(model, args.dense)
            if R.world == 1 and not args.no_vendor_baseline and L == 64:
                vendor_baseline = measure_vendor_baseline(B, args.dense)
    infer_bs16 = None
    if R.rank == 0 and R.world == 1 and not R.cpu and not args.no_extras and not args.no_extra_configs and L == 64 and B == 4:
        # the reference's own evaluation operating point: U-Net batch 16 = 8 prompts x classifier-free guidance
        # (configs/img_generation/sd-2-1_cc3m.yaml:47,50, scripts/metrics/generate_fid_images.py:104-128); same model, same mask
        try:
            infer_bs16 = measure_infer_batch(torch, ops, model, dev, 16, L, xdim, masked=not args.dense)
        except Exception as e:  # noqa: BLE001   (never lose the headline line to an extra)
            import traceback
            infer_bs16 = {"error": repr(e), "where": traceback.format_exc()[-500:]}
        torch.cuda.empty_cache()
    extra_configs = None
    if R.rank == 0 and R.world == 1 and not R.cpu and not args.no_extras and not args.no_extra_configs and L == 64:
        # BASELINE configs[2] and configs[4] in the same record (the driver runs only this default command): measured after
        # the headline, with the inference model released first
        del model, run
        if gout is not None:
            del graph, gout
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        extra_configs = {}
        for name, fn in (("train", run_train), ("finetune", run_finetune)):
            t0 = time.perf_counter()
            try:
                torch.cuda.reset_peak_memory_stats()
                rec = fn(R, steps=20, warmup=3)
                extra_configs[name] = {k: rec[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config",
                                                           "graph_nodes_per_step", "family_roofline", "max_mem_GiB", "loss") if k in rec}
                for k in ("launches_per_step", "launches_per_step_in_graphs", "collectives", "experts"):
                    if k in rec:
                        extra_configs[name][k] = rec[k]
            except Exception as e:  # noqa: BLE001   (never lose the headline line to an extra)
                import traceback
                extra_configs[name] = {"error": repr(e), "where": traceback.format_exc()[-600:]}
            extra_configs[name]["wall_s"] = round(time.perf_counter() - t0, 1)
            gc.collect()
            torch.cuda.empty_cache()
    if R.rank != 0:
        return None
    if infer_bs16 is not None:
        extra_configs = dict(extra_configs or {})
        extra_configs["infer_bs16"] = infer_bs16
    what = "mask=1" if args.dense else "fixed 50% mask (gated semantics)"
    line = {
        "metric": "denoise-steps/s (SD-2.1 U-Net forward, 512x512, bs=4 per GPU, " + ("dense)" if args.dense else "50% channel mask)"),
        "value": round(value, 3), "unit": "steps/s", "n_gpus": R.n_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: SD-2.1 UNet2DConditionModelGated.forward, 64x64 latents "
                               f"(512x512), bs={B}/GPU, {what}, seeded random-init weights, HIP graph replay",
                   "global_batch": B * R.world, "parallelism": f"replicas x{R.world} (no data-path collective)"},
        "per_gpu_steps_per_s": round(value / R.world, 3),
        "roofline": roofline, "roofline_groupnorm": roofline_gn, "roofline_attention": roofline_attn, "cpu_baseline": cpu_baseline,
        "gpu_vendor_baseline": vendor_baseline,
        "sustained": sustained, "extra_configs": extra_configs,
    }
    return line


def run_train(R: Rank, steps=None, warmup=None, extras: bool = True):
    """BASELINE configs[2] (N = 1) / configs[3] (N > 1): the APTP pruning train step, data-parallel."""
    torch, args, dev = R.torch, R.args, R.dev
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    from diffusion_pruning_amd import dist_utils
    from diffusion_pruning_amd.hypernet import HyperStructure
    from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
    from diffusion_pruning_amd.train_step import GraphedPrunerStep, PrunerStep, synthetic_batch
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated

    depth_order = [-1, -2, 0, 1, -3, -4, 2, 3, -5, -6, 4, 5, -7, 6]
    if R.cpu:
        unet, text_dim, n_e = _dryrun_install().install_train()
        latent = 8
    else:
        unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
        text_dim, n_e, latent = 768, 8, args.latent
    (unet.real if R.cpu else unet).freeze()
    st = unet.get_structure() if not R.cpu else unet.real.get_structure()
    torch.manual_seed(0)                                   # identical router replicas on every rank
    hn = HyperStructure(structure=st, input_dim=text_dim, wn_flag=False, linear_bias=True).to(dev)
    qz = StructureVectorQuantizer(n_e=n_e, structure=st, temperature=0.4, base=3, depth_order=depth_order,
                                  resource_aware_normalization=False, optimal_transport=True,
                                  fused_sinkhorn_allreduce=True).to(dev)
    hn.train(); qz.train()
    graphed = not (args.no_graph or R.cpu)
    step = (GraphedPrunerStep if graphed else PrunerStep)(unet, hn, qz)
    step.count_macs(latent)
    # capturable: with one rank the router itself (forward, chain rule, this optimizer) is replayed from HIP graphs too
    opt = torch.optim.AdamW(step.trainable_parameters(), lr=2e-4, **({} if R.cpu else {"capturable": True}))
    xdim = 1024 if not R.cpu else unet.real.config["cross_attention_dim"]
    batch = synthetic_batch(args.batch, latent, dev, seed=1234 + R.rank, cross_dim=xdim, text_dim=text_dim)   # rank-local shard
    nodes = fam_log = None
    if graphed:
        from diffusion_pruning_amd import graph_utils, ops
        graph_utils.KEEP_GRAPHS = True
        ops.LAUNCH_LOG = [] if (extras and R.rank == 0) else None
        try:
            step.capture(batch, optimizer=None if (args.eager_router or R.dist) else opt, pretrain=False)
        finally:
            graph_utils.KEEP_GRAPHS = False
            ops.LAUNCH_LOG = None
        nodes, fam_log = step.graph_nodes(), step._cap.get("launch_log")
        torch.cuda.reset_peak_memory_stats()
    timer = dist_utils.CollectiveTimer()
    out = {}

    def run():
        out["o"] = step.train_step(opt, batch, pretrain=R.cpu)

    elapsed = R.timed(run, steps, warmup)
    # collective share: a few more steps with the device-side stopwatch around every collective of the step
    dist_utils.COLLECTIVE_TIMER = timer
    n_probe = min(steps, 5)
    for _ in range(n_probe):
        run()
    R.sync()
    dist_utils.COLLECTIVE_TIMER = None
    spans = {k: round(v / n_probe, 4) for k, v in timer.total_ms().items()}
    coll_ms = sum(spans.values())
    # replicas must have stayed bit-identical (same averaged gradient on every rank)
    flat = torch.cat([p.detach().float().flatten() for p in step.trainable_parameters()])
    if R.world > 1:
        lo, hi = flat.clone(), flat.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        assert torch.equal(lo, hi), "router replicas diverged across ranks"
    assert torch.isfinite(flat).all()
    ms_per_step = elapsed / steps * 1e3
    value = R.world * steps / elapsed
    if R.rank != 0:
        return None
    o = out["o"]
    mem = None if R.cpu else round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    fam = None
    if graphed and extras and fam_log:
        from diffusion_pruning_amd import ops
        fam = family_roofline(ops, fam_log)
    return {
        "metric": "pruning-train-steps/s (APTP Pruner.step: router + dense teacher fwd + soft-masked student fwd/bwd, SD-2.1, "
                  f"{latent}x{latent} latents, bs={args.batch} per GPU; summed over data-parallel ranks)",
        "value": round(value, 3), "unit": "steps/s", "n_gpus": R.n_seen, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[3]" if R.world > 1 else "BASELINE configs[2]") +
                               f": APTP pruning train step, synthetic CC3M-shape batch, bs={args.batch}/GPU, "
                               + (("U-Net passes and the router (forward, chain rule, AdamW; Gumbel noise from the host generator) replayed from HIP graphs"
                                  if (graphed and step._cap.get("router")) else "U-Net passes replayed from HIP graphs, router eager") if graphed else "eager"),
                   "global_batch": args.batch * R.world,
                   "parallelism": f"dp{R.world}: frozen U-Net replicas; per step 1 fused all-gather [B,768+1620], 1 all-gather of "
                                  "Sinkhorn scores [B,8], 1 flat fp32 all-reduce of 1.26 M router gradients"},
        "per_gpu_steps_per_s": round(value / R.world, 3), "global_steps_per_s": round(steps / elapsed, 3),
        "samples_per_s": round(value * args.batch, 2),
        "graph_nodes_per_step": nodes,
        "launches_per_step_in_graphs": None if not nodes or None in nodes.values() else sum(nodes.values()),
        "family_roofline": fam,
        "collectives": {"ms_per_step": round(coll_ms, 4), "share_of_step": round(coll_ms / ms_per_step, 5), "spans_ms": spans,
                        "how": "HIP events on the launch stream around each collective, mean of %d steps" % n_probe},
        "loss": float(o["loss"].detach()), "resource_ratio": float(o["resource_ratio"]), "max_mem_GiB": mem,
        "replicas_identical_after_run": True,
    }


def expert_mask(structure, expert: int, device):
    """architecture code of benchmark expert 0..7 (BASELINE configs[4] "8 distinct arch codes, mixed width + depth pruning"):
    keep ratio 0.40 + 0.05 * expert of every width gate (seeded random positions), expert % 5 depth gates off"""
    import torch
    g = torch.Generator().manual_seed(1000 + expert)
    keep = 0.4 + 0.05 * expert
    width = []
    for sub in structure["width"]:
        for w in sub:
            m = torch.zeros(1, w)
            m[0, torch.randperm(w, generator=g)[:max(1, int(keep * w))]] = 0.9
            width.append(m.to(device))
    nd = sum(d for sub in structure["depth"] for d in sub)
    depth = [torch.full((1,), 0.9, device=device) for _ in range(nd)]
    for i in torch.randperm(nd, generator=g)[:expert % 5].tolist():
        depth[i] = torch.zeros(1, device=device)
    return {"width": width, "depth": depth}


def family_roofline(ops, launch_log):
    """MFMA roofline of the conv_gemm family INSIDE a captured training step: the launches recorded while the step was
    captured (forward and data-gradient contractions; weight gradients are a different kernel), replayed from a graph of
    their own between HIP events"""
    import torch
    if not launch_log:
        return None
    lib = ops._lib.load()
    stream = torch.cuda.Stream()

    def fn():
        s = torch.cuda.current_stream().cuda_stream
        for rec in launch_log:
            rc = getattr(lib, rec.get("fn", "aptp_conv_gemm"))(ctypes.byref(rec["params"]), s)
            assert rc == 0
    ms = _time_graph(torch, stream, fn, reps=5)
    flops = sum(r["flops"] for r in launch_log)
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "conv_gemm family (forward + data-gradient contractions of one step)", "launches": len(launch_log),
            "family_ms_per_step": round(ms, 3), "algorithmic_gflop_per_step": round(flops / 1e9, 1),
            "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4)}


def run_finetune(R: Rank, steps=None, warmup=None, extras: bool = True):
    """BASELINE configs[4]: expert fine-tune, one expert per GPU, no collective on the data path."""
    torch, args, dev = R.torch, R.args, R.dev
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    # --data-parallel: ALL ranks train the same expert; the packed gradients the replayed graph leaves behind are
    # averaged in buckets (reduce-scatter + all-gather) between the backward and the one-launch AdamW
    dp = bool(args.data_parallel) and not R.cpu
    expert = args.expert_offset % 8 if dp else (R.rank + args.expert_offset) % 8
    if R.cpu:
        step, batch, n_train, keep = _dryrun_install().install_finetune(expert, args.batch)
        nodes = fam = None
    else:
        from diffusion_pruning_amd import graph_utils, ops
        from diffusion_pruning_amd.train_step import GraphedFineTunerStep, synthetic_batch
        from diffusion_pruning_amd.unet import UNet2DConditionModelGated, UNet2DConditionModelPruned
        teacher = UNet2DConditionModelGated().init_synthetic(seed=0)
        student = UNet2DConditionModelPruned()
        student.load_state_dict(teacher.state_dict())
        teacher.to(dev).freeze()
        st = teacher.get_structure()
        teacher.set_structure(ones_mask(st, dev))
        student.to(dev)
        student.prune(expert_mask(st, expert, dev))
        batch = synthetic_batch(args.batch, args.latent, dev, seed=1234 + R.rank)
        step = GraphedFineTunerStep(student, teacher, lr=1e-5, data_parallel=dp and R.dist)
        graph_utils.KEEP_GRAPHS = True
        ops.LAUNCH_LOG = [] if (extras and R.rank == 0) else None
        try:
            step.capture(batch, offload_masters=True)
        finally:
            graph_utils.KEEP_GRAPHS = False
            ops.LAUNCH_LOG = None
        n_train, keep = step.trainer.n_trainable(), 0.4 + 0.05 * expert
        nodes = step.graph_nodes()
        fam = None
    out = {}

    def run():
        out["o"] = step.train_step(None, batch)

    torch.cuda.reset_peak_memory_stats() if not R.cpu else None
    elapsed = R.timed(run, steps, warmup)
    loss = float(out["o"]["loss"])
    assert loss == loss, "fine-tune loss is NaN"
    my_rate = steps / R.last_local_elapsed                      # this rank's own rate; `elapsed` is the MAX over ranks
    mem = None if R.cpu else round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    if not R.cpu and extras and R.rank == 0:
        from diffusion_pruning_amd import ops
        fam = family_roofline(ops, step._cap.get("launch_log"))
    # every rank's expert / parameter count / loss, gathered for the record (not on the data path)
    per_rank = [{"rank": R.rank, "expert": expert, "keep_ratio": round(keep, 2), "trainable_parameters": n_train, "loss": loss,
                 "steps_per_s": round(my_rate, 3), "max_mem_GiB": mem}]
    if R.dist:
        gathered = [None] * R.world
        torch.distributed.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    if R.rank != 0:
        return None
    ms_per_step = elapsed / steps * 1e3
    value = (1 if dp else R.world) * steps / elapsed             # data-parallel: ONE optimizer step per iteration for the whole job
    return {
        "metric": "expert-finetune-steps/s (APTP FineTuner.step: dense teacher fwd + pruned student fwd/bwd incl. weight gradients + "
                  "AdamW, SD-2.1, 64x64 latents, bs=4 per GPU; " + ("one expert on all GPUs, data-parallel)" if dp else
                                                                   "one expert per GPU, summed over ranks)"),
        "value": round(value, 3), "unit": "steps/s", "n_gpus": R.n_seen, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4]: expert fine-tune, 8 seeded architecture codes (keep 0.40-0.75, 0-4 depth gates "
                               f"off), " + (f"all ranks train expert {expert}" if dp else f"rank r trains expert (r + {args.expert_offset}) % 8")
                               + f", bs={args.batch}/GPU, step replayed from HIP graphs (teacher | student forward | loss + backward) "
                               "+ batched weight gradients + one-launch AdamW",
                   "global_batch": args.batch * R.world,
                   "parallelism": (f"dp{R.world}: one expert; the gradient arena is summed over the ranks IN PLACE in 256 MiB buckets "
                                   "(reduce-scatter + all-gather on a communication stream), AdamW follows bucket by bucket with "
                                   "grad_scale = 1/world" if dp else
                                   f"experts x{R.world}: one expert per GPU, no collective on the data path (finetune.py:27-28)")},
        "samples_per_s": round(args.batch * R.world * steps / elapsed, 2),
        "per_gpu_steps_per_s": round(value / (1 if dp else R.world), 3), "experts": per_rank, "graph_nodes_per_step": nodes,
        "launches_per_step": None if not nodes or None in nodes.values() else sum(nodes.values()) + 3,
        "family_roofline": fam, "max_mem_GiB": mem, "loss": loss,
        "gradient_exchange": None if (R.cpu or getattr(step, "reducer", None) is None) else {
            "buckets": len(step.reducer.buckets), "mode": step.reducer.mode, "arena_MiB": round(step.reducer.arena.numel() * 4 / 2 ** 20, 1),
            "collectives_per_step": step.reducer.stats["collectives"] / max(1, step.reducer.stats["steps"]),
            "per_tensor_ops_per_step": step.reducer.stats["tensor_ops"], "issued_on_backend": bool(step.reducer.active)},
    }


# ----------------------------------------------------------------------------------------------------------------------
# measurement legs (rank 0, after the timed region)
# ----------------------------------------------------------------------------------------------------------------------
def _time_graph(torch, stream, fn, reps=10):
    """device time (ms) of fn()'s launches: captured into a HIP graph on `stream`, replayed `reps` times between events"""
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            fn()
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            graph.replay()
        e1.record(stream)
        stream.synchronize()
        return e0.elapsed_time(e1) / reps


def _sclk_reader(local_rank: int):
    """current shader clock (MHz) from sysfs pp_dpm_sclk of the card this rank runs on -- a plain file read, no GPU API (None
    when the file is not there or not readable).  The in-kernel clock of an MFMA-dense loop reads up to ~10 % below it
    (MI355X_MICROARCH.md, DVFS give-back item 6): the figure is a companion of the rate, not a calibration."""
    import glob
    import re
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if not cards:
        return None
    # visible-device order is not the card order on a shared host: take the card whose file shows the HIGHEST current level
    # while this process loads its GPU (chosen at the first sample)
    state = {"path": None}

    def cur(path):
        try:
            with open(path) as f:
                for ln in f:
                    if "*" in ln:
                        m = re.search(r"(\d+)\s*Mhz", ln, re.I)
                        return int(m.group(1)) if m else None
        except OSError:
            return None
        return None

    def read():
        if state["path"] is None:
            best = max(((cur(p) or 0, p) for p in cards), default=(0, None))
            state["path"] = best[1]
            return best[0] or None
        return cur(state["path"])
    return read


def measure_sustained(replay, seconds: float, headline_value: float, local_rank: int):
    """>= `seconds` of back-to-back replays of the captured forward: steps/s over the whole span, per window of 100 steps
    (HIP events on the replay stream: slowest and median window), and the shader clock sampled by a host thread meanwhile."""
    import threading
    import torch
    n_win = max(3, int(seconds * max(headline_value, 1.0) / 100.0) + 1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
    reader = _sclk_reader(local_rank)
    clocks, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            v = reader()
            if v:
                clocks.append(v)
            stop.wait(0.05)
    th = None
    if reader is not None:
        th = threading.Thread(target=sample, daemon=True)
    for _ in range(100):
        replay()
    torch.cuda.synchronize()
    if th is not None:
        th.start()
    t0 = time.perf_counter()
    ev[0].record()
    for w in range(n_win):
        for _ in range(100):
            replay()
        ev[w + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    stop.set()
    if th is not None:
        th.join(timeout=1.0)
    wins = sorted(100.0 / (ev[i].elapsed_time(ev[i + 1]) * 1e-3) for i in range(n_win))
    sps = 100.0 * n_win / wall
    out = {"seconds": round(wall, 2), "steps": 100 * n_win, "steps_per_s": round(sps, 2),
           "window_100_steps": {"min_steps_per_s": round(wins[0], 2), "median_steps_per_s": round(wins[len(wins) // 2], 2),
                                "max_steps_per_s": round(wins[-1], 2)},
           "vs_headline": round(sps / headline_value, 4),
           "sclk_mhz": None if not clocks else {"min": min(clocks), "median": sorted(clocks)[len(clocks) // 2], "max": max(clocks),
                                                "samples": len(clocks), "source": "sysfs pp_dpm_sclk, 50 ms host-thread samples during the run"}}
    return out


def measure_infer_batch(torch, ops, model, dev, batch, latent, xdim, steps=10, warmup=2, masked=True):
    """the same forward at another U-Net batch, graph-replayed: steps/s, latents/s, the conv_gemm family's MFMA roofline inside it
    and how much of the family runs on the persistent stream-K macro-tiles (APTP_TILE_SK_*, csrc/conv_gemm_sk.hip)"""
    g = torch.Generator().manual_seed(4321)
    sample = torch.randn(batch, 4, latent, latent, generator=g).to(dev)
    ehs = torch.randn(batch, 77, xdim, generator=g).to(dev)
    t = torch.full((batch,), 500, dtype=torch.int64, device=dev)

    def step():
        return model(sample, t, ehs, return_dict=False)[0]
    with torch.no_grad():
        out = step()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gout = step()
        for _ in range(warmup):
            graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        assert torch.isfinite(gout).all()
        roof = measure_roofline(ops, step, dev, per_tile=True)
    sk = {k: v for k, v in roof["per_tile"].items() if k.isdigit() and int(k) >= ops.SK_TILE_FIRST}
    sk_ms = sum(v["ms"] for v in sk.values())
    flops = algorithmic_flops(masked, batch, latent) if latent == 64 else None
    del graph, gout
    return {"config": f"the headline model and mask at U-Net batch {batch} (the reference's evaluation batch: 8 prompts x CFG), HIP graph replay",
            "value": round(1e3 / ms, 3), "unit": "steps/s", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
            "latents_per_s": round(batch * 1e3 / ms, 1),
            "whole_step_tflops": None if flops is None else round(flops / (ms * 1e-3) / 1e12, 1),
            "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches_per_step", "family_ms_per_step",
                                              "algorithmic_gflop_per_step")},
            "stream_k": {"launches": sum(v["launches"] for v in sk.values()), "ms": round(sk_ms, 4),
                         "share_of_family_time": round(sk_ms / roof["family_ms_per_step"], 4) if roof["family_ms_per_step"] else None,
                         "per_tile": sk},
            "per_tile": roof["per_tile"]}


def measure_roofline(ops, step, dev, per_tile: bool = True):
    """Device time of every conv_gemm launch of one forward (the dominant kernel family), measured with HIP events on
    the launch stream around a HIP graph that replays exactly those launches; then the same per kernel instantiation."""
    import torch
    lib = ops._lib.load()
    ops.LAUNCH_LOG = []
    step()
    torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    stream = torch.cuda.Stream()

    def replayer(recs):
        def fn():
            s = torch.cuda.current_stream().cuda_stream
            for rec in recs:
                rc = getattr(lib, rec.get("fn", "aptp_conv_gemm"))(ctypes.byref(rec["params"]), s)
                assert rc == 0
        return fn
    total_ms = _time_graph(torch, stream, replayer(log))
    flops = sum(r["flops"] for r in log)
    n = len(log)
    achieved = flops / (total_ms * 1e-3) / 1e12
    per_tile_out = {}
    groups = {}
    for r in log:
        groups.setdefault(r["fn"] if "fn" in r else int(r["params"].tile), []).append(r)
    for tile, recs in (sorted(groups.items(), key=lambda kv: str(kv[0])) if per_tile else ()):
        ms = _time_graph(torch, stream, replayer(recs), reps=5)
        fl = sum(r["flops"] for r in recs)
        tf = fl / (ms * 1e-3) / 1e12
        per_tile_out[str(tile)] = {"launches": len(recs), "ms": round(ms, 4), "tflops": round(tf, 1), "frac": round(tf / PEAK_BF16_TFLOPS, 4)}
    # HBM-side bytes per launch of this kernel family come from separate rocprofv3 --pmc passes over this same command
    # (FETCH_SIZE and WRITE_SIZE cannot share a pass; gfx950 FETCH_SIZE correction applied) committed under profiles/.
    traffic, traffic_src = None, None
    import glob
    tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if tpaths:
        try:
            with open(tpaths[-1]) as fh:       # the latest round's passes (tools/pmc_traffic.py)
                traffic = round(json.load(fh)["hbm_side_bytes_per_launch"])
            traffic_src = ("profiles/" + os.path.basename(tpaths[-1]) +
                           " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py, bytes per launch)")
        except Exception:  # noqa: BLE001
            traffic = None
    return {"bound": "mfma", "kernel": "conv_gemm family (implicit-GEMM conv/linear, all tile instantiations, + aptp_ff_tail where it replaces three of them; per_tile keys = AptpTile ids of include/aptp_hip.h)",
            "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": n, "avg_launch_us": round(total_ms * 1e3 / n, 2),
            "family_ms_per_step": round(total_ms, 4), "algorithmic_gflop_per_step": round(flops / 1e9, 1), "per_tile": per_tile_out}


def measure_gn_roofline(ops, step, dev):
    """HBM roofline of the GroupNorm(+SiLU) launches of one forward: algorithmic bytes = x read once + y written once."""
    import torch
    lib = ops._lib.load()
    ops.GN_LAUNCH_LOG = []
    step()
    torch.cuda.synchronize()
    log, ops.GN_LAUNCH_LOG = ops.GN_LAUNCH_LOG, None
    if not log:
        return None
    stream = torch.cuda.Stream()

    def fn():
        s = torch.cuda.current_stream().cuda_stream
        for rec in log:
            rc = lib.aptp_groupnorm(ctypes.byref(rec["params"]), s)
            assert rc == 0
    ms = _time_graph(torch, stream, fn)
    nbytes = sum(r["bytes"] for r in log)
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "aptp_groupnorm family (gn_group / gn_stats / gn_finalize* / gn_apply)", "achieved": round(gbs, 1),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
            "calls_per_step": len(log), "family_ms_per_step": round(ms, 4), "algorithmic_MB_per_step": round(nbytes / 1e6, 1)}


def measure_attn_roofline(ops, step, dev):
    """MFMA roofline of the attention launches of one forward (aptp_attention: flash-style, head_dim 64), replayed from a HIP graph on
    their recorded operands; self-attention (Lk = Lq) and cross-attention (77 keys) apart."""
    import torch
    lib = ops._lib.load()
    ops.ATTN_LAUNCH_LOG = []
    step()
    torch.cuda.synchronize()
    log, ops.ATTN_LAUNCH_LOG = ops.ATTN_LAUNCH_LOG, None
    if not log:
        return None
    stream = torch.cuda.Stream()

    def timed(recs):
        def fn():
            s = torch.cuda.current_stream().cuda_stream
            for rec in recs:
                rc = lib.aptp_attention(ctypes.byref(rec["params"]), s)
                assert rc == 0
        ms = _time_graph(torch, stream, fn)
        fl = sum(r["flops"] for r in recs)
        tf = fl / (ms * 1e-3) / 1e12
        return {"launches": len(recs), "ms": round(ms, 4), "algorithmic_gflop": round(fl / 1e9, 1), "tflops": round(tf, 1), "frac": round(tf / PEAK_BF16_TFLOPS, 4)}
    allr = timed(log)
    out = {"bound": "mfma", "kernel": "aptp_attention (attn_fwd_sp_kernel on whole 128-blocks, attn_fwd_kernel<NG> otherwise)", "achieved": allr["tflops"],
           "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": allr["frac"], "traffic": None, "calls_per_step": allr["launches"],
           "family_ms_per_step": allr["ms"], "algorithmic_gflop_per_step": allr["algorithmic_gflop"]}
    self_a = [r for r in log if r["Lk"] == r["params"].Lq]
    cross = [r for r in log if r["Lk"] != r["params"].Lq]
    if self_a:
        out["self_attention"] = timed(self_a)
        big = [r for r in self_a if r["Lk"] >= 4096]
        if big:
            out["self_attention_level64"] = timed(big)
    if cross:
        out["cross_attention"] = timed(cross)
    return out


def measure_peaks(ops, dev):
    """Denominators measured on this box: a plain 8192^3 bf16 GEMM through hipBLASLt (torch.matmul) and
    through this repo's own kernel, and a 1 GiB device-to-device copy (read + write bytes / time)."""
    import torch
    n = 8192
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)

    def t(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    def t_graph(fn, reps):
        """both GEMMs the way the step runs its launches: replayed from a HIP graph (no host work between launches)"""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(5):
                fn()
        return t(g.replay, reps) / 5
    fl = 2.0 * n ** 3
    lib_tf = fl / t_graph(lambda: torch.matmul(a, b), 4) / 1e12
    pw = ops.pack_weight(b.t().float().contiguous(), None, device=dev)
    x = a.view(1, n, 1, n)
    own_tf = fl / t_graph(lambda: ops.conv_gemm(x, pw, pad=0), 4) / 1e12
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    gbs = 2.0 * src.numel() / t(lambda: dst.copy_(src), 10) / 1e9
    return {"gemm_bf16_8192_hipblaslt_tflops": round(lib_tf, 1), "gemm_bf16_8192_own_kernel_tflops": round(own_tf, 1),
            "stream_copy_1GiB_GBs": round(gbs, 1)}


def measure_vendor_baseline(batch, dense, budget_s: float = 150.0):
    """Same-node vendor-library yardstick (stated baseline, not the target): tools/bench_vendor.py in a child process -- the oracle's
    op sequence on this GPU in bf16 through torch-ROCm's MIOpen / hipBLASLt / SDPA, gated semantics (dense compute + mask
    multiply, as the reference's UNet2DConditionModelGated does), HIP-graph replay.  Bounded: the child is given `budget_s`."""
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "bench_vendor.py"), "--batch", str(batch)]
    if dense:
        cmd.append("--dense")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s)
        last = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not last:
            return {"error": (r.stderr or r.stdout)[-400:]}
        return json.loads(last[-1])
    except subprocess.TimeoutExpired:
        return {"error": f"not finished within {budget_s:.0f} s (library kernel selection on a fresh box)"}
    except Exception as e:  # noqa: BLE001   (never lose the headline line to a baseline)
        return {"error": repr(e)}


def measure_cpu_baseline(model, dense):
    """The oracle (PyTorch CPU fp32 restatement of the reference's diffusers path) on this host: median of 5 denoise steps
    of the headline workload (bs=4) after one warm-up, and the same for BASELINE configs[0] (bs=1, dense)."""
    import torch
    from oracle import unet_oracle as O
    cfg = O.SD21
    params = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    # torch's CPU kernels stop scaling (and regress badly) far below the 256 hardware threads of the GPU host;
    # 32 threads is what the sample is timed with and what "cores" reports.
    cores = min(os.cpu_count() or 1, int(os.environ.get("APTP_CPU_THREADS", "32")))
    torch.set_num_threads(cores)

    def med(mask, bs, reps=5):
        gates = O.assign_gates(cfg, mask)
        sample, t, ehs = O.synthetic_inputs(cfg, bs, 64)
        O.unet_forward(params, cfg, sample, t, ehs, gates, "gated")            # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            O.unet_forward(params, cfg, sample, t, ehs, gates, "gated")
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), ts
    with torch.no_grad():
        dt1, ts1 = med(O.ones_mask(cfg), 1)
        dt4, ts4 = med(O.ones_mask(cfg) if dense else O.fixed_half_mask(cfg), 4)
    return {"value": round(1.0 / dt4, 5), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"median of 5 denoise steps at bs=4 (the full step of the headline workload) after one warm-up: {dt4:.2f} s/step "
                      f"(runs {', '.join('%.2f' % v for v in ts4)}), gated semantics (dense compute + mask multiply, as the reference "
                      f"does), fp32, torch {torch.__version__} CPU, {cores} threads",
            "config0_bs1_dense": {"value": round(1.0 / dt1, 5), "unit": "steps/s", "s_per_step": round(dt1, 3),
                                  "sample": "BASELINE configs[0]: bs=1 dense, median of 5 after one warm-up (runs "
                                            + ", ".join("%.2f" % v for v in ts1) + ")"}}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    R = Rank(args)
    line = {"train": run_train, "finetune": run_finetune, "infer": run_infer}[args.config](R)
    if line is not None:
        if R.cpu:
            line["data"] = "DRYRUN on CPU with emulated ops and a tiny model: exercises launch/rendezvous/reporting only, NOT a measurement"
        print(json.dumps(line), flush=True)
    R.barrier()                                   # ranks leave together (rank 0 ran the measurement legs meanwhile)
    R.finish()


if __name__ == "__main__":
    main()
