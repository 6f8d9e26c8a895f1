"""The N > 1 gradient-exchange path (sgdm_amd/ddp.py) on CPU: world_size 2 over gloo, 127.0.0.1 rendezvous."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sgdm_amd.ddp import BucketReducer, GradArena
    shapes = [("out.2.weight", (3, 8, 3, 3)), ("out.2.bias", (3,)), ("mid.weight", (64, 64, 3, 3)),
              ("unused.weight", (5, 7)), ("in.weight", (8, 3, 3, 3)), ("in.bias", (8,))]
    arena = GradArena(shapes, "cpu", bucket_bytes=64 * 1024)
    assert len(arena.buckets) >= 2                      # the 147 KB tensor forces a cut
    red = BucketReducer(arena)
    g = torch.Generator().manual_seed(100 + rank)
    local = {}
    red.start()
    for name, shape in shapes:                           # "backward" produces gradients in arena order
        if name == "unused.weight":
            continue                                     # never written: stays zero on every rank
        local[name] = torch.randn(shape, generator=g)
        arena.grad(name).copy_(local[name])
        bi = arena.bucket_of[name]
        if name == arena.buckets[bi][2]:
            red.bucket_ready(bi)                         # overlapped send as soon as the bucket is complete
    red.finish()
    # expected: mean over ranks of the per-rank tensors
    exp = {}
    for name, shape in shapes:
        if name == "unused.weight":
            exp[name] = torch.zeros(shape)
            continue
        acc = torch.zeros(shape)
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            for n2, s2 in shapes:
                if n2 == "unused.weight":
                    continue
                t = torch.randn(s2, generator=gr)
                if n2 == name:
                    acc += t
        exp[name] = acc / world
    ok = all(torch.allclose(arena.grad(n), exp[n], atol=1e-6) for n, _ in shapes)
    q.put((rank, ok, len(arena.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_arena_views_are_aligned_and_ordered():
    from sgdm_amd.ddp import GradArena
    shapes = [("a", (3,)), ("b", (5, 5)), ("c", (2, 2, 2))]
    ar = GradArena(shapes, "cpu", bucket_bytes=64)
    for n, s in shapes:
        v = ar.grad(n)
        assert tuple(v.shape) == s and v.data_ptr() % 16 == 0
        assert v.data_ptr() >= ar.flat.data_ptr()
    assert [ar.bucket_of[n] for n, _ in shapes] == sorted(ar.bucket_of[n] for n, _ in shapes)
    ar.grad("b").fill_(2.0)
    assert float(ar.flat.sum()) == 50.0


def test_torch_ddp_wrapper_detection_and_exclusion():
    """sgdm_amd.ddp.find_torch_ddp_wrapper / exclude_from_torch_ddp on CPU modules over a single-rank gloo group: the
    wrapper around a PARENT of the module is found; a module marked with exclude_from_torch_ddp is skipped by torch's
    reducer (DDP's own _ddp_params_and_buffers_to_ignore contract)"""
    import os
    import tempfile
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from sgdm_amd.ddp import exclude_from_torch_ddp, find_torch_ddp_wrapper
    with tempfile.TemporaryDirectory() as td:
        dist.init_process_group("gloo", init_method=f"file://{os.path.join(td, 'store')}", rank=0, world_size=1)
        try:
            class Holder(torch.nn.Module):
                def __init__(self):
                    super().__init__()
                    self.dynamic = torch.nn.Linear(4, 4)
                    self.other = torch.nn.Linear(4, 2)
                    self.dynamic.register_buffer("shadow", torch.zeros(3))

                def forward(self, x):
                    return self.other(self.dynamic(x))
            free = Holder()
            assert find_torch_ddp_wrapper(free.dynamic) is None
            root = Holder()
            names = exclude_from_torch_ddp(root, root.dynamic)
            assert sorted(names) == ["dynamic.bias", "dynamic.shadow", "dynamic.weight"]
            w = DDP(root)
            assert find_torch_ddp_wrapper(root.dynamic) is w and find_torch_ddp_wrapper(free.dynamic) is None
            assert w.parameters_to_ignore == set(names)
        finally:
            dist.destroy_process_group()
