"""world_size-2 gloo tests (CPU) of the N>1 path: distributed Sinkhorn routing (reference all-reduce form and the fused
single-all-gather form), cross-rank gather with local autograd, fused router-gradient all-reduce, and a data-parallel
PrunerStep around a stub U-Net."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import unet_oracle as O

DEPTH_ORDER = [-1, -2, 0, 1, -3, -4, 2, 3, -5, -6, 4, 5, -7, 6]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, world=2):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r
    return {r[0]: {k: (torch.from_numpy(v) if hasattr(v, "dtype") and hasattr(v, "shape") else v) for k, v in r[2].items()}
            for r in res}


def _entry(fn, rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        out = fn(rank, world)
        dist.destroy_process_group()
        # tensors -> numpy (plain pickles; torch's shared-memory fd passing breaks when the child exits first)
        out = {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}
        q.put((rank, "ok", out))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "err", traceback.format_exc()))


def _quantizer(fused):
    from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
    torch.manual_seed(0)
    q = StructureVectorQuantizer(n_e=4, structure=O.get_structure(O.TINY), temperature=0.4, base=3, depth_order=DEPTH_ORDER,
                                 resource_aware_normalization=False, optimal_transport=True, fused_sinkhorn_allreduce=fused)
    return q


def _sinkhorn_worker(rank, world):
    out = {}
    torch.manual_seed(100)
    scores = torch.rand(8, 4)                       # global [B=8, K=4] similarity block, identical on every rank
    local = scores[rank * 4:(rank + 1) * 4].clone()
    for fused in (False, True):
        q = _quantizer(fused)
        Q = q._sinkhorn_fused(local) if fused else q._sinkhorn(local.clone(), True)
        out["fused" if fused else "ref"] = Q
    return out


def test_distributed_sinkhorn_equals_global_sinkhorn():
    res = _run(_sinkhorn_worker)
    torch.manual_seed(100)
    scores = torch.rand(8, 4)
    q = _quantizer(False)
    # single-process Sinkhorn on the concatenated batch (quantizer.py:308-330 with B = global batch)
    full = q._sinkhorn(scores.clone(), False)
    for r in (0, 1):
        assert torch.allclose(res[r]["ref"], full[r * 4:(r + 1) * 4], atol=1e-6)
        assert torch.allclose(res[r]["fused"], full[r * 4:(r + 1) * 4], atol=1e-6)


def _gather_worker(rank, world):
    from diffusion_pruning_amd.train_step import allreduce_mean_grads, gather_with_local_grad
    a = (torch.arange(6.0).view(2, 3) + 10 * rank).requires_grad_()
    b = (torch.arange(4.0).view(2, 2) + 100 * rank).requires_grad_()
    ga, gb = gather_with_local_grad(a, b)
    (ga.sum() * 2 + gb.sum() * 3).backward()
    lin = nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.fill_(1.0); lin.bias.fill_(0.0)
    lin(torch.full((1, 3), float(rank + 1))).sum().backward()
    allreduce_mean_grads(lin.parameters())
    return {"ga": ga.detach(), "gb": gb.detach(), "a_grad": a.grad, "b_grad": b.grad, "w_grad": lin.weight.grad, "b_grad2": lin.bias.grad}


def test_gather_with_local_grad_and_fused_grad_allreduce():
    res = _run(_gather_worker)
    for r in (0, 1):
        ga = torch.cat([torch.arange(6.0).view(2, 3), torch.arange(6.0).view(2, 3) + 10])
        assert torch.equal(res[r]["ga"], ga)
        assert res[r]["gb"].shape == (4, 2)
        assert torch.equal(res[r]["a_grad"], torch.full((2, 3), 2.0))       # only the local block carries gradient
        assert torch.equal(res[r]["b_grad"], torch.full((2, 2), 3.0))
        assert torch.allclose(res[r]["w_grad"], torch.full((2, 3), 1.5))    # mean of rank grads 1 and 2
        assert torch.allclose(res[r]["b_grad2"], torch.ones(2))


class StubUNet(nn.Module):
    """reference-API stand-in whose output depends differentiably on the gates (the real U-Net needs the GPU)"""

    def __init__(self, real):
        super().__init__()
        self.real = real                                     # product model on CPU: structure plumbing + MAC accounting
        self.down_blocks = nn.ModuleList([nn.Identity() for _ in range(4)])
        self.mid_block = nn.Identity()
        self.up_blocks = nn.ModuleList([nn.Identity() for _ in range(4)])

    def set_structure(self, sep):
        w, d = list(sep["width"]), list(sep["depth"])
        self.scale = torch.cat([x.mean(dim=1, keepdim=True) for x in w] + [x[:, None] for x in d], dim=1).mean(dim=1)
        self.real.set_structure({"width": w, "depth": d})

    def forward(self, sample, t, ehs):
        out = sample * self.scale.view(-1, 1, 1, 1)
        for b in self.down_blocks:
            b((out, None))
        self.mid_block(out)
        for b in self.up_blocks:
            b(out)

        class R:
            pass
        r = R()
        r.sample = out
        return r

    def calc_macs(self):
        return self.real.calc_macs()

    def count_macs(self, n):
        return self.real.count_macs(n)

    prunable_macs_list = property(lambda self: self.real.prunable_macs_list)
    resource_info_dict = property(lambda self: self.real.resource_info_dict)


def _step_worker(rank, world):
    from diffusion_pruning_amd.hypernet import HyperStructure
    from diffusion_pruning_amd.train_step import PrunerStep, synthetic_batch
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    cfg = O.TINY
    real = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                     cross_attention_dim=cfg.cross_attention_dim)
    torch.manual_seed(1)                                      # identical replicas on every rank
    hn = HyperStructure(structure=real.get_structure(), input_dim=16, wn_flag=False, linear_bias=True)
    qz = _quantizer(True)
    step = PrunerStep(StubUNet(real), hn, qz)
    hn.train(); qz.train()
    step.count_macs(8)
    opt = torch.optim.SGD(step.trainable_parameters(), lr=0.1)
    batch = synthetic_batch(2, 8, "cpu", seed=50 + rank, cross_dim=cfg.cross_attention_dim, text_dim=16)   # rank-local shard
    torch.manual_seed(7 + rank)
    out = step.train_step(opt, batch, pretrain=True)
    flat = torch.cat([p.detach().flatten() for p in step.trainable_parameters()])
    return {"loss": float(out["loss"]), "params": flat}


def test_data_parallel_pruner_step_keeps_replicas_in_sync():
    res = _run(_step_worker)
    assert torch.isfinite(res[0]["params"]).all()
    assert torch.equal(res[0]["params"], res[1]["params"])    # same averaged gradient applied on both ranks
    assert res[0]["loss"] != res[1]["loss"]                   # different data shards


def _bucket_worker(rank, world, mode="all_reduce"):
    """BucketedGradReducer: buckets in reverse registration order, launched from post-accumulate hooks,
    mean written back into .grad; a parameter that received no gradient still takes part (zeros)."""
    from diffusion_pruning_amd.train_step import BucketedGradReducer
    torch.manual_seed(3)
    net = nn.Sequential(nn.Linear(8, 16), nn.Linear(16, 16), nn.Linear(16, 4))
    unused = nn.Parameter(torch.ones(5))
    params = list(net.parameters()) + [unused]
    red = BucketedGradReducer(params, bucket_bytes=300, wire_dtype=torch.float32, mode=mode)      # several small buckets
    nb = len(red.buckets)
    x = torch.full((2, 8), float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    red.finish()
    out = {"nb": nb, "unused": unused.grad.clone()}
    for i, (p, g) in enumerate(zip(net.parameters(), local)):
        out[f"g{i}"] = p.grad.clone()
        out[f"l{i}"] = g
    # second step through the same reducer (buffers reused, counters reset)
    for p in params:
        p.grad = None
    net(x * 2).sum().backward()
    red.finish()
    out["g0_step2"] = net[0].weight.grad.clone()
    out["l0_step2_scale"] = 2.0
    return out


def test_bucketed_grad_reducer_means_gradients_across_ranks():
    res = _run(_bucket_worker)
    assert res[0]["nb"] >= 3
    n = sum(1 for k in res[0] if k.startswith("g") and k[1:].isdigit())
    for i in range(n):
        mean = (res[0][f"l{i}"] + res[1][f"l{i}"]) / 2
        for r in (0, 1):
            assert torch.allclose(res[r][f"g{i}"], mean, atol=1e-6), i
    assert torch.equal(res[0]["unused"], torch.zeros(5))
    assert torch.equal(res[0]["g0_step2"], res[1]["g0_step2"])


def _bucket_worker_rs_ag(rank, world):
    return _bucket_worker(rank, world, mode="rs_ag")


def test_bucketed_grad_reducer_reduce_scatter_all_gather_mode():
    """mode "rs_ag" (the direct full-mesh exchange on xGMI): reduce-scatter + all-gather per bucket, bucket sizes
    that are not multiples of the world size (padded shards), same means as the all-reduce form"""
    res = _run(_bucket_worker_rs_ag)
    assert res[0]["nb"] >= 3
    n = sum(1 for k in res[0] if k.startswith("g") and k[1:].isdigit())
    for i in range(n):
        mean = (res[0][f"l{i}"] + res[1][f"l{i}"]) / 2
        for r in (0, 1):
            assert torch.allclose(res[r][f"g{i}"], mean, atol=1e-6), i
    assert torch.equal(res[0]["unused"], torch.zeros(5))
    assert torch.equal(res[0]["g0_step2"], res[1]["g0_step2"])


def _exchange_all_worker(rank, world):
    """hooks=False + exchange_all(): the form GraphedFineTunerStep(data_parallel=True) uses on the gradients a replayed
    graph leaves behind -- means written IN PLACE (the optimizer's table holds the addresses), odd sizes, both modes"""
    from diffusion_pruning_amd.train_step import BucketedGradReducer
    out = {}
    for mode in ("all_reduce", "rs_ag"):
        torch.manual_seed(5)
        ps = [nn.Parameter(torch.randn(s)) for s in ((7, 3), (5,), (64, 9), (1,))]
        for p in ps:
            p.grad = torch.full_like(p, float(rank + 1)) * p.detach()
        ptrs = [p.grad.data_ptr() for p in ps]
        red = BucketedGradReducer(ps, bucket_bytes=200, wire_dtype=torch.float32, mode=mode, hooks=False)
        red.exchange_all()
        red.exchange_all()                         # a second exchange of already equal gradients changes nothing
        out[mode + "_inplace"] = all(p.grad.data_ptr() == q for p, q in zip(ps, ptrs))
        out[mode] = torch.cat([(p.grad / p.detach()).flatten() for p in ps])
    return out


def test_exchange_all_without_hooks_writes_means_in_place():
    res = _run(_exchange_all_worker)
    for mode in ("all_reduce", "rs_ag"):
        for r in (0, 1):
            assert res[r][mode + "_inplace"]
            assert torch.allclose(res[r][mode], torch.full_like(res[r][mode], 1.5), atol=1e-6)


def _arena_worker(rank, world):
    """ArenaGradReducer over a flat fp32 'gradient arena': in-place bucketed exchange, both modes"""
    from diffusion_pruning_amd.train_step import ArenaGradReducer
    out = {}
    n = 64 * world * 37                       # several buckets of 64 * world * 8 elements, a short last one
    for mode in ("rs_ag", "all_reduce"):
        g = torch.Generator().manual_seed(7 + rank)
        arena = torch.randn(n, generator=g)
        mine = arena.clone()
        red = ArenaGradReducer(arena, bucket_bytes=4 * 64 * world * 8, mode=mode)
        seen = []
        ptr = arena.data_ptr()
        red.exchange(lambda i: seen.append(i))
        assert arena.data_ptr() == ptr and seen == list(range(len(red.buckets)))
        out[mode] = arena.clone()
        out[mode + "_mine"] = mine
        out[mode + "_nb"] = len(red.buckets)
        out[mode + "_coll"] = red.stats["collectives"]
        out[mode + "_ops"] = red.stats["tensor_ops"]
        # bucket_of: a range is complete with the bucket that holds its last element
        assert red.bucket_of(0, 1) == 0 and red.bucket_of(0, n) == len(red.buckets) - 1
        a, b = red.bounds[1]
        assert red.bucket_of(a - 3, 3) == 0 and red.bucket_of(a - 3, 4) == 1
    return out


def test_arena_reducer_sums_in_place_with_two_collectives_per_bucket():
    """train_step.ArenaGradReducer (the graphed fine-tune's data-parallel exchange, trainer.py:1616): the arena ends up holding the
    SUM over the ranks, bit-equal on both ranks and equal to the directly computed sum; 2 collectives per bucket in rs_ag mode,
    1 in all_reduce mode, no per-tensor operation, nothing re-allocated"""
    res = _run(_arena_worker)
    for mode, per in (("rs_ag", 2), ("all_reduce", 1)):
        total = res[0][mode + "_mine"] + res[1][mode + "_mine"]
        assert torch.equal(res[0][mode], res[1][mode])                      # replicas agree bit for bit
        assert torch.equal(res[0][mode], total)                             # two-term fp32 sums: exact
        assert res[0][mode + "_nb"] == 5
        assert res[0][mode + "_coll"] == per * res[0][mode + "_nb"] and res[0][mode + "_ops"] == 0
