"""The floor of a chain of dependent launches in a replayed HIP graph on this box: N trivial kernels (one-element add), N kernels
that each read + write a 10 MB tensor (an elementwise pass over a level-64 activation), captured on one stream.  Against the
forward's 339 launches in 5.26 ms (15.5 us per launch) this says how much of the per-launch latency is the runtime's
launch-to-launch cost and how much is inside the kernels."""
import torch

dev = torch.device("cuda:0")
N = 340
x1 = torch.zeros(1, device=dev)
xb = torch.zeros(16384, 320, device=dev, dtype=torch.bfloat16)


def timed_graph(fn, reps=50):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def tiny():
    for _ in range(N):
        x1.add_(1.0)


def stream_pass():
    for _ in range(N):
        xb.add_(1.0)


t1 = timed_graph(tiny)
t2 = timed_graph(stream_pass)
print(f"{N} dependent one-element kernels: {t1:.3f} ms per replay = {t1 / N * 1e3:.2f} us per launch")
print(f"{N} dependent 10 MB read+write passes: {t2:.3f} ms per replay = {t2 / N * 1e3:.2f} us per launch "
      f"({2 * xb.numel() * 2 / (t2 / N * 1e-3) / 1e12:.2f} TB/s)")
