"""Experiment (round 4): does the headline forward (bs = 4) finish sooner as TWO half-batch chains running next to each other?

Most launches of the forward are bound by their dependent chain (launch -> first operand -> K loop -> epilogue), not by the chip's
throughput, so two independent chains of half-size launches on two hardware queues might overlap each other's fixed costs.
Forms timed (same weights, same mask, same inputs; every output compared with the one-graph result):
  one      one graph, batch 4                                   (what bench.py measures)
  half     one graph, batch 2                                   (for reference: the latency-bound floor of a chain)
  forked   one graph: samples 0-1 on the capture stream, 2-3 on a forked stream
  two      two graphs of batch 2 replayed on two streams that really run concurrently (graph_utils.concurrent_stream)

usage: python3 tools/proto/bench_split_batch.py [out.txt]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffusion_pruning_amd import ops  # noqa: E402
from diffusion_pruning_amd.graph_utils import concurrent_stream  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated  # noqa: E402


def timed(run, reps=60, warm=10):
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
    st = model.get_structure()
    model.set_structure(bench.fixed_half_mask(st, dev))
    g = torch.Generator(device="cpu").manual_seed(1234)
    B = 4
    sample = torch.randn(B, 4, 64, 64, generator=g).to(dev)
    ehs = torch.randn(B, 77, model.config["cross_attention_dim"], generator=g).to(dev)
    t = torch.full((B,), 500, dtype=torch.int64, device=dev)
    lines = []

    def fwd(lo, hi):
        return model(sample[lo:hi].contiguous(), t[lo:hi].contiguous(), ehs[lo:hi].contiguous(), return_dict=False)[0]

    def capture(fn):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = fn()
        return gr, out

    with torch.no_grad():
        fwd(0, 4); fwd(0, 2)
        torch.cuda.synchronize()
        g_one, o_one = capture(lambda: fwd(0, 4))
        ms_one = timed(g_one.replay)
        lines.append(f"one     (1 graph, bs 4)            {ms_one:7.3f} ms  {1e3 / ms_one:7.2f} steps/s")
        ref = o_one.float().clone()

        g_half, o_half = capture(lambda: fwd(0, 2))
        ms_half = timed(g_half.replay)
        d = float((o_half.float() - ref[:2]).abs().max())
        lines.append(f"half    (1 graph, bs 2)            {ms_half:7.3f} ms  (max |diff| to one[:2] {d:.2e})")

        def forked():
            main_s = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                with ops.scratch_domain("half1"):
                    o1 = fwd(2, 4)
            o0 = fwd(0, 2)
            main_s.wait_stream(side)
            return o0, o1
        try:
            g_fork, (f0, f1) = capture(forked)
            ms_fork = timed(g_fork.replay)
            d = float((torch.cat([f0, f1]).float() - ref).abs().max())
            lines.append(f"forked  (1 graph, 2 branches)      {ms_fork:7.3f} ms  {1e3 / ms_fork:7.2f} steps/s  (max |diff| {d:.2e})")
        except Exception as e:  # noqa: BLE001
            lines.append(f"forked  FAILED: {type(e).__name__}: {str(e)[:200]}")

        probe = []
        side = concurrent_stream(log=probe)
        g_a, o_a = capture(lambda: fwd(0, 2))
        with ops.scratch_domain("half1"):
            g_b, o_b = capture(lambda: fwd(2, 4))
        main_s = torch.cuda.current_stream()

        def two():
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                g_b.replay()
            g_a.replay()
            main_s.wait_stream(side)
        ms_two = timed(two)
        d = float((torch.cat([o_a, o_b]).float() - ref).abs().max())
        lines.append(f"two     (2 graphs, 2 streams)      {ms_two:7.3f} ms  {1e3 / ms_two:7.2f} steps/s  (max |diff| {d:.2e}; stream probe {probe})")

        def serial():
            g_a.replay(); g_b.replay()
        ms_ser = timed(serial)
        lines.append(f"serial  (2 graphs, 1 stream)       {ms_ser:7.3f} ms")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
