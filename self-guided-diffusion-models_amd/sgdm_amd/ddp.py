"""Data-parallel gradient exchange for the HIP training step (replaces ``pl.trainer.strategy=ddp``,
reference config/pl/default.yaml:2, README.md:84-94).

One process per GPU (``torch.distributed``, backend "nccl" == RCCL on ROCm).  The backward program writes every
parameter gradient into ONE flat arena laid out in production order (last layers first), cut into buckets.
As soon as the launches that fill a bucket have been issued, an event is recorded on the compute stream and the
bucket's all-reduce is enqueued on a side stream, so the exchange over xGMI runs under the rest of the backward.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few LARGE buckets (default 64 MB, 301 MB of gradients
=> 5 collectives) keep every link busy without the fixed per-collective cost of torch DDP's 25 MB default.

Unlike torch DDP nothing is broadcast per step: EMA shadows and schedule tables are rank-deterministic
(SURVEY.md 2.3 "drop").  The arena holds exactly the parameters the static backward program produces, in the
same order on every rank, so all ranks reduce identical byte ranges; parameters off the path
(``to_cond_tokens_2d.*`` of unetca_fast) are not in it and keep ``grad = None`` like in the reference
(which needs ``ddp_find_unused_parameters_true`` for them, README.md:90-94).

Device-agnostic on purpose: the same code runs on CPU tensors over gloo in tests/test_ddp_gloo.py.
"""
import torch
import torch.distributed as dist


class GradArena:
    """flat gradient storage + bucket bookkeeping"""

    def __init__(self, shapes, device, bucket_bytes=64 << 20, dtype=torch.float32):
        """shapes: ordered [(name, shape)] in the order the backward produces the gradients"""
        self.names = [n for n, _ in shapes]
        self.views, self.offsets = {}, {}
        total = 0
        for name, shape in shapes:
            numel = 1
            for s in shape:
                numel *= int(s)
            self.offsets[name] = (total, numel)
            total += (numel + 3) // 4 * 4                    # keep every view 16-byte aligned
        self.flat = torch.zeros(max(total, 4), dtype=dtype, device=device)
        for name, shape in shapes:
            off, numel = self.offsets[name]
            self.views[name] = self.flat[off:off + numel].view(*shape)
        per = max(1, bucket_bytes // self.flat.element_size())
        self.buckets = []                                     # (start, end, last parameter name inside)
        start = 0
        for name in self.names:
            off, numel = self.offsets[name]
            end = off + (numel + 3) // 4 * 4
            if end - start >= per:
                self.buckets.append((start, end, name))
                start = end
        if start < total or not self.buckets:
            self.buckets.append((start, max(total, 4), self.names[-1] if self.names else None))
        self.bucket_of = {}
        for bi, (s, e, _) in enumerate(self.buckets):
            for name in self.names:
                off, _n = self.offsets[name]
                if s <= off < e:
                    self.bucket_of[name] = bi

    def grad(self, name):
        return self.views[name]


class BucketReducer:
    """overlapped all-reduce(avg) of the arena's buckets on a side stream"""

    def __init__(self, arena, group=None, average=True):
        """average=False: the producer already scaled its gradients by 1/world (folded into the
        un-scaling of the backward program), the collective is a plain SUM"""
        self.arena, self.group, self.average = arena, group, average
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.cuda = arena.flat.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.pending = []
        self.done = set()

    def start(self):
        self.pending, self.done = [], set()

    def bucket_ready(self, bi):
        """call right after the last launch writing into bucket ``bi`` has been issued on the current stream"""
        if self.world == 1 or bi in self.done:
            return
        self.done.add(bi)
        s, e, _ = self.arena.buckets[bi]
        chunk = self.arena.flat[s:e]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()                                       # on the compute stream
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.pending.append((w, chunk))
        else:
            w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((w, chunk))

    def finish(self):
        """flush the buckets not yet sent, wait for all collectives, average"""
        if self.world == 1:
            return
        for bi in range(len(self.arena.buckets)):
            self.bucket_ready(bi)
        for w, _ in self.pending:
            w.wait()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        if self.average:
            self.arena.flat.mul_(1.0 / self.world)
        self.pending = []


# ------------------------------------------------------------------------------------------------------------------
# coexistence with torch's DistributedDataParallel (the reference's unchanged launch line: pl.trainer.strategy=ddp,
# README.md:84-94, config/pl/default.yaml:2)
# ------------------------------------------------------------------------------------------------------------------
def find_torch_ddp_wrapper(module):
    """the DistributedDataParallel instance whose wrapped module tree contains `module`, or None.  The drop-in UNet sits
    inside the LightningModule that PL's DDP strategy wraps, so it cannot see the wrapper from its own attributes: one
    scan of the live DDP objects (at the first training backward, never again)."""
    import gc
    from torch.nn.parallel import DistributedDataParallel as DDP
    for obj in gc.get_objects():
        try:
            if isinstance(obj, DDP) and any(m is module for m in obj.module.modules()):
                return obj
        except ReferenceError:
            continue
    return None


def exclude_from_torch_ddp(root, *modules):
    """call BEFORE wrapping `root` in DistributedDataParallel: torch DDP then neither reduces the gradients nor broadcasts
    the buffers of `modules` (the HIP UNet, its LitEma) -- their exchange is this file's bucketed all-reduce inside the
    backward program, and EMA shadows / schedule tables are rank-deterministic.  Uses DDP's own
    `_ddp_params_and_buffers_to_ignore` contract (fully qualified names under `root`)."""
    ids = set()
    for m in modules:
        ids.update(id(t) for t in m.parameters())
        ids.update(id(t) for t in m.buffers())
    names = [n for n, t in list(root.named_parameters()) + list(root.named_buffers()) if id(t) in ids]
    prev = list(getattr(root, "_ddp_params_and_buffers_to_ignore", []))
    root._ddp_params_and_buffers_to_ignore = prev + [n for n in names if n not in prev]
    return names


def torch_ddp_ignores(wrapper, module):
    """True iff torch's DistributedDataParallel `wrapper` ignores EVERY parameter of `module` (by tensor identity: names
    of other submodules may share a suffix with ours)"""
    ignored = set(getattr(wrapper, "parameters_to_ignore", None)
                  or getattr(wrapper.module, "_ddp_params_and_buffers_to_ignore", []) or [])
    if not ignored:
        return False
    ignored_ids = {id(t) for n, t in wrapper.module.named_parameters() if n in ignored}
    return all(id(p) in ignored_ids for p in module.parameters())


# ------------------------------------------------------------------------------------------------------------------
# what torch DDP does at construction and this path must do itself: identical replicas before the first step
# ------------------------------------------------------------------------------------------------------------------
def sync_initial_state(module, group=None, src=0):
    """broadcast rank `src`'s parameters and buffers to every rank, ONCE per module (torch DDP's construction-time
    `_sync_module_states`; the reference relies on it through pl.trainer.strategy=ddp, config/pl/default.yaml:2).
    In-place copies that bump the tensors' versions, so packed weights follow.  Returns the number of tensors sent."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    if getattr(module, "_hip_ddp_synced", False):
        return 0
    tensors = list(module.parameters()) + list(module.buffers())
    # one flat buffer per dtype: a few large broadcasts instead of ~400 small ones.  The copies back are in-place writes on
    # the parameters themselves (under no_grad): they bump the tensors' versions, which is what the packed-weight caches key on
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for dtype, ts in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
    module._hip_ddp_synced = True
    return len(tensors)


def reserved_cus(model=None):
    """compute units the persistent conv kernels leave free during a data-parallel TRAINING step.

    The conv kernel runs one block per CU with all of the CU's registers; RCCL's all-reduce kernels on the side stream
    need CUs of their own.  Without a reserve the two fight launch by launch (tests/test_hip_contention.py: a launch
    whose blocks do not all fit takes up to 2x).  With world == 1, or the exchange off, nothing is reserved.
    `SGDM_RESERVE_CUS` (default 16 = two per XCD; RCCL's workgroups are dealt round-robin over the XCDs like everyone
    else's) -- pair it with NCCL_MAX_NCHANNELS <= the reserve (bench.py does)."""
    import os
    forced = getattr(model, "hip_reserve_cus", None) if model is not None else None
    if forced is not None:                           # tests / tuning: a reserve without a process group
        return max(0, int(forced))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    if model is not None and not getattr(model, "hip_ddp", True):
        return 0
    return max(0, int(os.environ.get("SGDM_RESERVE_CUS", "16")))
