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


def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from sgdm_amd.ddp import BucketReducer, GradArena
    # the production order of a small unetca_fast-like backward: head first, FiLM stages interleaved, embedding MLPs last;
    # `to_cond_tokens_2d.*` (off the path, README.md:90-94) is NOT in the arena on any rank
    shapes = [("out.2.weight", (3, 32, 3, 3)), ("out.2.bias", (3,)), ("output_blocks.2.0.out_layers.3.weight", (32, 32, 3, 3)),
              ("output_blocks.2.0.emb_layers.1.weight", (64, 128)), ("output_blocks.2.0.emb_layers.1.bias", (64,)),
              ("middle_block.0.in_layers.2.weight", (64, 64, 3, 3)), ("middle_block.0.emb_layers.1.weight", (128, 128)),
              ("input_blocks.1.0.in_layers.2.weight", (32, 32, 3, 3)), ("input_blocks.1.0.emb_layers.1.weight", (64, 128)),
              ("input_blocks.0.0.weight", (32, 3, 3, 3)), ("time_embed.2.weight", (128, 128)), ("time_embed.0.weight", (128, 32)),
              ("time_embed.0.bias", (128,))]
    arena = GradArena(shapes, "cpu", bucket_bytes=96 * 1024, tail_bytes=48 * 1024)
    red = BucketReducer(arena, average=False)             # the product: SUM, 1 / world folded into the producers
    unused = torch.full((5, 7), 3.25)                     # a parameter gradient outside the arena stays what it was
    red.start()
    g = torch.Generator().manual_seed(500 + rank)
    last = len(arena.buckets) - 1
    sent_from_hook = []
    for name, shape in shapes:
        arena.grad(name).copy_(torch.randn(shape, generator=g) / world)
        bi = arena.bucket_of[name]
        if name == arena.buckets[bi][2] and bi != last:  # (train.Backward.wrote: the last bucket waits for the health flag)
            red.bucket_ready(bi)
            sent_from_hook.append(bi)
    arena.health.fill_(1.0 if rank == 3 else 0.0)         # rank 3's balanced tail timed out
    red.finish()
    exp = {name: torch.zeros(shape) for name, shape in shapes}
    for r in range(world):
        gr = torch.Generator().manual_seed(500 + r)
        for name, shape in shapes:
            exp[name] += torch.randn(shape, generator=gr) / world
    ok = all(torch.allclose(arena.grad(n), exp[n], atol=1e-6) for n, _ in shapes)
    import hashlib
    digest = hashlib.sha256(arena.flat.numpy().tobytes()).hexdigest()
    q.put((rank, ok, tuple(arena.buckets), tuple(sorted(arena.offsets.items())), float(arena.health[0]), digest,
           float(unused.sum()), tuple(sent_from_hook), arena.flat.numel()))
    dist.barrier()
    dist.destroy_process_group()


def test_arena_buckets_and_health_flag_world8_gloo():
    """what an 8-GPU node would exercise, minus the wire -- eight ranks cut the SAME byte ranges out of
    the same production order (bucket table and offsets identical on every rank), reduce them to bit-identical sums, leave a
    gradient that is not on the path alone, cap the last bucket (the only one no launch hides) and carry the step's health flag in
    its tail slot: one rank's time-out reaches every rank, so every rank's optimizer skips and every rank raises together."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[1] for r in res), [r[:2] for r in res]
    assert len({r[2] for r in res}) == 1 and len({r[3] for r in res}) == 1          # same buckets, same offsets
    assert len({r[5] for r in res}) == 1                                            # bit-identical arenas after the exchange
    assert all(r[4] == 1.0 for r in res)                                            # SUM of the flags: everyone knows
    assert all(r[6] == 3.25 * 35 for r in res)
    buckets, numel = res[0][2], res[0][8]
    assert len(buckets) >= 3 and buckets[0][0] == 0 and buckets[-1][1] == numel
    assert all(a[1] == b[0] for a, b in zip(buckets, buckets[1:]))                   # contiguous, no gaps, no overlap
    assert (buckets[-1][1] - buckets[-1][0]) * 4 <= 48 * 1024                        # capped tail
    assert res[0][7] == tuple(range(len(buckets) - 1))                              # every bucket but the last left from its hook


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


# ------------------------------------------------------------------------------------------------------------------
# round 4: what torch DDP's constructor did and the HIP exchange has to do itself
# ------------------------------------------------------------------------------------------------------------------
def _sync_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sgdm_amd.ddp import sync_initial_state
    torch.manual_seed(1000 + rank)                       # deliberately DIFFERENT initial weights per rank
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.GroupNorm(4, 8), torch.nn.Linear(8, 5))
    net.register_buffer("shadow", torch.randn(7))
    net.register_buffer("num_updates", torch.tensor(rank, dtype=torch.int32))
    before = [p.detach().clone() for p in net.parameters()]
    versions = [p._version for p in net.parameters()]
    sent = sync_initial_state(net)
    again = sync_initial_state(net)                      # once per module
    torch.manual_seed(1000)                              # what rank 0 drew
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.GroupNorm(4, 8), torch.nn.Linear(8, 5))
    ref_shadow = torch.randn(7)
    same = all(torch.equal(a, b) for a, b in zip(net.parameters(), ref.parameters()))
    same = same and torch.equal(net.shadow, ref_shadow) and int(net.num_updates) == 0
    changed = any(not torch.equal(a, b.detach()) for a, b in zip(before, net.parameters()))
    bumped = all(p._version > v for p, v in zip(net.parameters(), versions))      # packed-weight caches key on versions
    q.put((rank, same, changed, bumped, sent, again))
    dist.barrier()
    dist.destroy_process_group()


def test_initial_state_broadcast_world2_gloo():
    """ranks that start from different weights end up with rank 0's parameters AND buffers (bit for bit), the copies
    bump the parameter versions (so packed weights are rebuilt), and the broadcast happens once per module"""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, changed, bumped, sent, again in res:
        assert same and bumped and sent == 8 and again == 0, res
        assert changed == (rank != 0), res


def test_pl_strategy_hooks():
    """HipDDPStrategy on a stand-in for Lightning's DDPStrategy that drives the hooks the way both generations do
    (1.6-1.9: configure_ddp wraps LightningDistributedModule(model) through _setup_model; 2.x: _setup_model(model) and
    _register_ddp_hooks asserting a DistributedDataParallel instance): the module must come out bare either way."""
    from sgdm_amd import pl_strategy

    class Wrapped:
        def __init__(self, m):
            self.module = m

    calls = []

    class FakeDDPStrategy:
        def __init__(self, model, generation):
            self.model, self.generation = model, generation

        def _setup_model(self, model):                   # Lightning: DistributedDataParallel(module=model, ...)
            calls.append("base._setup_model")
            return Wrapped(model)

        def _register_ddp_hooks(self):
            calls.append("base._register_ddp_hooks")
            assert isinstance(self.model, Wrapped)       # 2.x asserts the wrapper type on CUDA

        def configure_ddp(self):
            inner = self.model if self.generation == 2 else ("LightningDistributedModule", self.model)
            self.model = self._setup_model(inner)
            self._register_ddp_hooks()

        def setup(self):                                 # trainer.fit(): strategy.setup() -> configure_ddp()
            self.configure_ddp()

    cls = pl_strategy.make_strategy(FakeDDPStrategy)
    assert cls.strategy_name == "hip_ddp"
    for generation in (1, 2):
        net = torch.nn.Linear(3, 2)
        st = cls(net, generation)
        st.setup()
        assert st.model is net and not calls, (generation, calls)
        assert st._setup_model(net) is net
    # without Lightning the public name raises on use, importing the module does not
    if pl_strategy._Base is None:
        import pytest
        with pytest.raises(ImportError):
            pl_strategy.HipDDPStrategy()


def test_pl_strategy_step_dispatch():
    """Lightning 1.6-1.9's DDPStrategy.training_step is `self.model(*args)` and relies on the
    LightningDistributedModule wrapper to redirect forward -> training_step.  HipDDPStrategy keeps the bare module, so it
    must dispatch the four step methods itself, under the precision plugin's context."""
    import contextlib
    from sgdm_amd import pl_strategy
    entered = []

    class Plugin:
        def _ctx(self, name):
            @contextlib.contextmanager
            def cm():
                entered.append(name)
                yield
            return cm

        def __getattr__(self, name):
            if name.endswith("_context"):
                return self._ctx(name)
            raise AttributeError(name)

    class LM(torch.nn.Module):
        def forward(self, *a, **k):
            raise AssertionError("forward() reached: the step was not redirected")

        def training_step(self, batch, batch_idx):
            return ("train", batch, batch_idx)

        def validation_step(self, batch, batch_idx):
            return ("val", batch, batch_idx)

        def test_step(self, batch, batch_idx):
            return ("test", batch, batch_idx)

        def predict_step(self, batch, batch_idx):
            return ("predict", batch, batch_idx)

    class OldDDPStrategy:                                # the 1.6-1.9 shape of the base class
        def __init__(self, model):
            self.model, self.precision_plugin = model, Plugin()
            self.env_calls = 0

        @property
        def lightning_module(self):
            return getattr(self.model, "module", self.model)

        def setup_environment(self):
            self.env_calls += 1

        def training_step(self, *args, **kwargs):
            return self.model(*args, **kwargs)

        validation_step = test_step = predict_step = training_step

    cls = pl_strategy.make_strategy(OldDDPStrategy)
    st = cls(LM())
    assert st.training_step("b", 3) == ("train", "b", 3)
    assert st.validation_step("b", 4) == ("val", "b", 4)
    assert st.test_step("b", 5) == ("test", "b", 5)
    assert st.predict_step("b", 6) == ("predict", "b", 6)
    assert entered == ["train_step_context", "val_step_context", "test_step_context", "predict_step_context"]
    # setup_environment: RCCL's channel cap is in the environment before the base class creates the process group
    old = os.environ.pop("NCCL_MAX_NCHANNELS", None)
    try:
        st.setup_environment()
        assert st.env_calls == 1 and os.environ.get("NCCL_MAX_NCHANNELS") == os.environ.get("SGDM_RESERVE_CUS", "16")
        os.environ["NCCL_MAX_NCHANNELS"] = "4"           # the user's own setting wins
        st.setup_environment()
        assert os.environ["NCCL_MAX_NCHANNELS"] == "4"
    finally:
        os.environ.pop("NCCL_MAX_NCHANNELS", None)
        if old is not None:
            os.environ["NCCL_MAX_NCHANNELS"] = old


def test_torch_ddp_ignores_is_by_identity():
    """a parameter of ANOTHER submodule whose name ends like one of ours must not count as ignored"""
    from sgdm_amd.ddp import torch_ddp_ignores

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dynamic = torch.nn.Linear(4, 4)
            self.ema_dynamic = torch.nn.Linear(4, 4)     # "ema_dynamic.weight".endswith("dynamic.weight") style clash

    class FakeWrapper:
        def __init__(self, root, names):
            self.module, self.parameters_to_ignore = root, set(names)

    root = Holder()
    assert not torch_ddp_ignores(FakeWrapper(root, ["ema_dynamic.weight", "ema_dynamic.bias"]), root.dynamic)
    assert not torch_ddp_ignores(FakeWrapper(root, ["dynamic.weight"]), root.dynamic)          # only some of them
    assert torch_ddp_ignores(FakeWrapper(root, ["dynamic.weight", "dynamic.bias"]), root.dynamic)
    assert not torch_ddp_ignores(FakeWrapper(root, []), root.dynamic)


def test_reserved_cus_policy(monkeypatch):
    """one rank reserves nothing; the attribute override works without a process group; a backend that launches no
    kernels on the device (gloo) needs no compute units"""
    import tempfile
    from sgdm_amd.ddp import exchange_active, reserved_cus
    assert reserved_cus(None) == 0
    m = torch.nn.Linear(1, 1)
    assert reserved_cus(m) == 0
    m.hip_reserve_cus = 32
    assert reserved_cus(m) == 32
    del m.hip_reserve_cus
    with tempfile.TemporaryDirectory() as td:
        dist.init_process_group("gloo", init_method=f"file://{os.path.join(td, 'store')}", rank=0, world_size=1)
        try:
            assert not exchange_active(m) and reserved_cus(m) == 0
            m.hip_force_exchange = True                  # one rank, exchange forced: active, but gloo reserves nothing
            assert exchange_active(m) and reserved_cus(m) == 0
            m.hip_ddp = False
            assert not exchange_active(m)
        finally:
            dist.destroy_process_group()


def _forced_worker(q):
    """one-rank group with the exchange forced: the collectives are issued, marks recorded, values unchanged"""
    import tempfile
    from sgdm_amd.ddp import BucketReducer, GradArena
    with tempfile.TemporaryDirectory() as td:
        dist.init_process_group("gloo", init_method=f"file://{os.path.join(td, 'store')}", rank=0, world_size=1)
        shapes = [("c.weight", (64, 64, 3, 3)), ("b.weight", (64, 64, 3, 3)), ("a.weight", (8, 3))]
        arena = GradArena(shapes, "cpu", bucket_bytes=64 * 1024)
        idle = BucketReducer(arena)
        forced = BucketReducer(arena, force=True)
        assert not idle.active and forced.active and forced.world == 1
        g = torch.Generator().manual_seed(1)
        vals = {n: torch.randn(s, generator=g) for n, s in shapes}
        forced.start()
        for n, _ in shapes:
            arena.grad(n).copy_(vals[n])
            bi = arena.bucket_of[n]
            if n == arena.buckets[bi][2]:
                forced.bucket_ready(bi)
        forced.backward_done()
        forced.finish()
        st = forced.overlap_stats()
        ok = all(torch.equal(arena.grad(n), vals[n]) for n, _ in shapes)
        q.put((ok, st))
        dist.destroy_process_group()


def test_forced_exchange_one_rank_records_overlap():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(q,))
    p.start()
    ok, st = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0 and ok
    assert st["buckets"] == 3 and len(st["per_bucket"]) == 3
    assert st["exchange_ms"] >= 0 and st["exposed_exchange_ms"] >= 0 and st["backward_ms"] > 0
    assert 0.0 <= st["first_bucket_at_frac_of_backward"] <= 1.0
    # buckets are enqueued in production order and each completes after it was enqueued
    enq = [b["enqueued_at_ms"] for b in st["per_bucket"]]
    assert enq == sorted(enq) and all(b["complete_at_ms"] >= b["enqueued_at_ms"] for b in st["per_bucket"])
