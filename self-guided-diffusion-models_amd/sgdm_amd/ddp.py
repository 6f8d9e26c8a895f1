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
import os
import time
import warnings

import torch
import torch.distributed as dist


def _initialized():
    return dist.is_available() and dist.is_initialized()


def exchange_forced(model=None):
    """the exchange runs even with ONE rank (``model.hip_force_exchange = True`` or ``SGDM_FORCE_EXCHANGE=1``): every
    collective, stream dependency and CU reserve of the multi-GPU step executes on the hardware at hand -- a
    world-size-1 ``nccl`` process group puts librccl, ProcessGroupNCCL's internal stream and the side-stream event
    ordering below through a real run on one GPU (tests/test_hip_rccl_world1.py)"""
    return bool(getattr(model, "hip_force_exchange", False)) or os.environ.get("SGDM_FORCE_EXCHANGE", "0") == "1"


def exchange_active(model=None):
    """True iff the backward program of `model` exchanges its gradients itself: a process group exists, the native
    exchange is not switched off (``model.hip_ddp = False``: torch DDP owns it) and there is somebody to exchange with
    (or the exchange is forced)"""
    if not _initialized():
        return False
    if model is not None and not getattr(model, "hip_ddp", True):
        return False
    return dist.get_world_size() > 1 or exchange_forced(model)


def backend_runs_on_gpu(group=None):
    """does the process group's backend launch kernels on this device ("nccl" == RCCL)?  gloo moves the buckets through
    host memory: it needs no compute units, so nothing is reserved for it"""
    if not _initialized():
        return False
    try:
        return "nccl" in str(dist.get_backend(group)).lower()
    except Exception:
        return False


class GradArena:
    """flat gradient storage + bucket bookkeeping"""

    def __init__(self, shapes, device, bucket_bytes=64 << 20, dtype=torch.float32, tail_bytes=16 << 20):
        """shapes: ordered [(name, shape)] in the order the backward produces the gradients.
        tail_bytes: cap of the LAST bucket.  It is complete only with the backward's last launch, so its collective is the
        part of the exchange no launch hides: whatever the bucket size, the last one is cut so that it holds at most this
        much (as long as a parameter boundary allows)."""
        self.names = [n for n, _ in shapes]
        self.views, self.offsets = {}, {}
        total = 0
        for name, shape in shapes:
            numel = 1
            for s in shape:
                numel *= int(s)
            self.offsets[name] = (total, numel)
            total += (numel + 3) // 4 * 4                    # keep every view 16-byte aligned
        # one more quad behind the last gradient: [0] = health flag of the step (train.Backward.run writes 0 / 1 before the
        # last bucket is sent; after the SUM every rank knows whether ANY rank's gradients are invalid)
        self.health_off = total
        total += 4
        self.flat = torch.zeros(max(total, 4), dtype=dtype, device=device)
        self.health = self.flat[self.health_off:self.health_off + 1]
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
        if start < total or not self.buckets:          # (always: the health quad lies behind the last gradient)
            self.buckets.append((start, max(total, 4), self.names[-1] if self.names else None))
        # cap the last bucket: split it at the first parameter boundary from which the rest fits into tail_bytes
        cap = max(4, tail_bytes // self.flat.element_size())
        ls, le, ln = self.buckets[-1]
        if le - ls > cap:
            prev = None
            for name in self.names:
                off, numel = self.offsets[name]
                if off > ls and le - off <= cap:
                    self.buckets[-1:] = [(ls, off, prev), (off, le, ln)]
                    break
                if off >= ls:
                    prev = name
        self.bucket_of = {}
        for bi, (s, e, _) in enumerate(self.buckets):
            for name in self.names:
                off, _n = self.offsets[name]
                if s <= off < e:
                    self.bucket_of[name] = bi

    def grad(self, name):
        return self.views[name]


class BucketReducer:
    """overlapped all-reduce(avg) of the arena's buckets on a side stream.

    Every step also leaves a record of WHEN the exchange ran relative to the backward program (`overlap_stats`): a mark
    when the backward starts, one per bucket when its collective is enqueued behind the launches that fill it, one per
    bucket when the collective has completed, and one when the last backward launch has been issued.  On the GPU the
    marks are HIP events on the streams the work runs on (compute stream / side stream), on CPU tensors host clocks."""

    def __init__(self, arena, group=None, average=True, force=False):
        """average=False: the producer already scaled its gradients by 1/world (folded into the
        un-scaling of the backward program), the collective is a plain SUM.
        force=True: issue the collectives even in a one-rank group (exchange_forced)"""
        self.arena, self.group, self.average = arena, group, average
        self.world = dist.get_world_size(group) if _initialized() else 1
        self.active = self.world > 1 or (bool(force) and _initialized())
        self.cuda = arena.flat.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        # RCCL: work.wait() is a dependency of the current STREAM on the collective's end, the host does not wait -- the
        # completion mark can be enqueued right behind the collective.  gloo: wait() blocks the host, so it moves to finish()
        self.stream_ordered = self.cuda and backend_runs_on_gpu(group)
        self.pending = []
        self.done = set()
        self.marks, self.t_start, self.t_bwd_end = [], None, None

    def _mark(self):
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                                       # on the current stream
            return ev
        return time.perf_counter()

    def start(self):
        self.pending, self.done = [], set()
        self.marks, self.t_bwd_end = [], None
        self.t_start = self._mark() if self.active else None

    def backward_done(self):
        """call when the last launch of the backward program has been issued (before finish)"""
        if self.active:
            self.t_bwd_end = self._mark()

    def bucket_ready(self, bi):
        """call right after the last launch writing into bucket ``bi`` has been issued on the current stream"""
        if not self.active or bi in self.done:
            return
        self.done.add(bi)
        s, e, _ = self.arena.buckets[bi]
        chunk = self.arena.flat[s:e]
        if self.cuda:
            ev = self._mark()                                 # on the compute stream: the bucket is complete here
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                rec = [bi, ev, None]
                if self.stream_ordered:
                    w.wait()                                  # side stream <- the collective's end (no host wait)
                    rec[2] = self._mark()
                self.marks.append(rec)
                self.pending.append((w, chunk, rec))
        else:
            rec = [bi, self._mark(), None]
            w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.marks.append(rec)
            self.pending.append((w, chunk, rec))

    def finish(self):
        """flush the buckets not yet sent, wait for all collectives, average"""
        if not self.active:
            return
        if self.t_bwd_end is None:
            self.backward_done()
        for bi in range(len(self.arena.buckets)):
            self.bucket_ready(bi)
        for w, _, rec in self.pending:
            if rec[2] is None:
                if self.cuda:
                    with torch.cuda.stream(self.stream):
                        w.wait()
                        rec[2] = self._mark()
                else:
                    w.wait()
                    rec[2] = self._mark()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        if self.average and self.world > 1:
            self.arena.flat[:self.arena.health_off].mul_(1.0 / self.world)
        self.pending = []

    def overlap_stats(self):
        """timing of the LAST step's exchange relative to its backward program (synchronises the device):
        exchange_ms = first collective enqueued -> last collective complete; exposed_exchange_ms = what of that lies
        behind the end of the backward program (the part no launch hides); first_bucket_at_frac_of_backward = where in the
        backward the first collective could start (0 = at its first launch, 1 = only at its end: no overlap at all)"""
        if not self.marks or self.t_start is None or self.t_bwd_end is None or any(r[2] is None for r in self.marks):
            return None
        if self.cuda:
            torch.cuda.synchronize()
            el = lambda a, b: float(a.elapsed_time(b))
        else:
            el = lambda a, b: 1000.0 * (b - a)
        bwd = el(self.t_start, self.t_bwd_end)
        enq = [el(self.t_start, r[1]) for r in self.marks]
        end = [el(self.t_start, r[2]) for r in self.marks]
        esz = self.arena.flat.element_size()
        late = sum((self.arena.buckets[r[0]][1] - self.arena.buckets[r[0]][0]) * esz for r, q in zip(self.marks, enq) if q > 0.9 * bwd)

        def modelled_exposed(gbps):
            """what of the exchange would lie behind the end of the backward if every collective ran at `gbps` GB/s of
            all-reduced bytes, one after the other on the side stream, each starting when its bucket is enqueued (as
            measured) and the one before it has finished: the figure an 8-GPU run would show, from a one-rank record"""
            t = 0.0
            for r, q in zip(self.marks, enq):
                nbytes = (self.arena.buckets[r[0]][1] - self.arena.buckets[r[0]][0]) * esz
                t = max(t, q) + nbytes / (gbps * 1e9) * 1e3
            return round(max(0.0, t - bwd), 3)
        return dict(buckets=len(self.marks), backward_ms=round(bwd, 3), mbytes_enqueued_after_0p9_of_backward=round(late / 2 ** 20, 1),
                    modelled_exposed_ms_at_50GBps=modelled_exposed(50.0), modelled_exposed_ms_at_100GBps=modelled_exposed(100.0),
                    exchange_ms=round(max(end) - min(enq), 3),
                    exposed_exchange_ms=round(max(0.0, max(end) - bwd), 3),
                    first_bucket_at_frac_of_backward=round(min(enq) / bwd, 4) if bwd > 0 else None,
                    per_bucket=[dict(bucket=r[0], mbytes=round((self.arena.buckets[r[0]][1] - self.arena.buckets[r[0]][0]) * esz / 2 ** 20, 1),
                                     enqueued_at_ms=round(q, 3), complete_at_ms=round(d, 3))
                                for r, q, d in zip(self.marks, enq, end)])


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
def sync_initial_state(module, group=None, src=0, late=False):
    """broadcast rank `src`'s parameters and buffers to every rank, ONCE per module (torch DDP's construction-time
    `_sync_module_states`; the reference relies on it through pl.trainer.strategy=ddp, config/pl/default.yaml:2).
    In-place copies that bump the tensors' versions, so packed weights follow.  Returns the number of tensors sent.

    Collective: every rank must call it at the same point.  Call it on the ROOT module (LightningModule: UNet + LitEma
    shadows) BEFORE anything is derived from the parameters.  `late=True` is the safety net inside the first training
    forward (train.forward_train), which can only see the UNet: a rank whose values the broadcast CHANGED warns that
    whatever was built from the old values (EMA shadows, optimizer state loaded per rank) is stale."""
    if not _initialized():
        return 0
    if dist.get_world_size(group) == 1 and not exchange_forced(module):
        return 0
    if getattr(module, "_hip_ddp_synced", False):
        return 0
    if backend_runs_on_gpu(group) and exchange_group() is None:
        # RCCL's half of the CU reserve.  Normally the buckets travel on a communicator of the exchange's own whose
        # workgroup count is capped (exchange_group, created -- collectively, all ranks are here -- just above): nothing
        # process-wide is touched and the user's other communicators (validation gathers, other models) keep RCCL's
        # defaults.  Only a torch / RCCL build without the per-communicator option falls back to the
        # environment variable, before what is then the first collective of the job.  The job owners that want the
        # process-wide cap as well set it themselves: HipDDPStrategy.setup_environment, bench.py.
        cap_exchange_channels()
    tensors = list(module.parameters()) + list(module.buffers())
    # one flat buffer per dtype: a few large broadcasts instead of ~400 small ones.  The copies back are in-place writes on
    # the parameters themselves (under no_grad): they bump the tensors' versions, which is what the packed-weight caches key on
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    changed = False
    with torch.no_grad():
        for dtype, ts in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            mine = flat.clone() if late else None
            dist.broadcast(flat, src=src, group=group)
            if late and not torch.equal(mine.view(torch.uint8), flat.view(torch.uint8)):
                changed = True
            del mine
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
    for sub in module.modules():                     # a submodule handed in later (the UNet inside the LightningModule)
        sub._hip_ddp_synced = True                   # is not sent a second time
    if changed:
        warnings.warn(f"sgdm_amd: rank {dist.get_rank()} started from parameters that differ from rank {src}'s; they were "
                      "overwritten at the first training forward.  Objects built from the old values before this point "
                      "(LitEma shadows, a FusedAdamWEma's EMA copy) are stale on this rank: call "
                      "sgdm_amd.ddp.sync_initial_state(root_module) right after the process group is created, before "
                      "building them (HipDDPStrategy and bench.py do)")
    return len(tensors)


def reserve_setting():
    """`SGDM_RESERVE_CUS` (default 16 = two compute units per XCD; RCCL's workgroups are dealt round-robin over the XCDs
    like everyone else's)"""
    return max(0, int(os.environ.get("SGDM_RESERVE_CUS", "16")))


def reserved_cus(model=None):
    """compute units the persistent conv kernels leave free during a data-parallel TRAINING step.

    The conv kernel runs one block per CU with all of the CU's registers; RCCL's all-reduce kernels on the side stream
    need CUs of their own.  Without a reserve the two fight launch by launch (tests/test_hip_contention.py: a launch
    whose blocks do not all fit takes up to 2x).  Nothing is reserved when nothing is exchanged (one rank and no forced
    exchange, or the native exchange off) or when the backend launches no kernels on this device (gloo).  The other half
    of the contract -- RCCL may not take MORE than the reserve -- is `cap_exchange_channels` / `exchange_group`."""
    forced = getattr(model, "hip_reserve_cus", None) if model is not None else None
    if forced is not None:                           # tests / tuning: a reserve without a process group
        return max(0, int(forced))
    if not exchange_active(model) or not backend_runs_on_gpu():
        return 0
    return reserve_setting()


def communicator_exists(group=None):
    """best effort: has the RCCL communicator of `group` (default group) been created already?  (ProcessGroupNCCL creates
    it lazily at the first collective; RCCL reads NCCL_* parameters when the first communicator of the PROCESS is made)"""
    if not _initialized() or not backend_runs_on_gpu(group):
        return False
    try:
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        be = pg._get_backend(torch.device("cuda"))
        return bool(be._is_initialized())
    except Exception:
        return False


def cap_exchange_channels(reserve=None):
    """RCCL's half of the CU reserve: NCCL_MAX_NCHANNELS <= the compute units the backward program leaves free, set in the
    product (not in a launcher script) and only if the user has not set it.  RCCL reads the variable when the process
    creates its first communicator, so this runs BEFORE the process group's first collective: from
    `HipDDPStrategy.setup_environment` (before Lightning initialises the group), from `sync_initial_state` (the first
    collective of a plain torch.distributed job) and from bench.py.  Returns the value in force; warns when a
    communicator already exists and the variable was not in its environment -- then only `exchange_group`'s
    per-communicator cap protects the reserve."""
    reserve = reserve_setting() if reserve is None else int(reserve)
    if reserve <= 0:
        return None
    cur = os.environ.get("NCCL_MAX_NCHANNELS")
    if cur is not None:
        if int(cur) > reserve:
            warnings.warn(f"sgdm_amd: NCCL_MAX_NCHANNELS={cur} exceeds the {reserve} compute units the backward program "
                          "reserves for the gradient exchange (SGDM_RESERVE_CUS): RCCL and the conv kernels will compete")
        return int(cur)
    if communicator_exists():
        warnings.warn("sgdm_amd: an RCCL communicator exists already; NCCL_MAX_NCHANNELS set now does not reach it. The "
                      "gradient exchange runs on a communicator of its own with a per-communicator cap instead "
                      "(sgdm_amd.ddp.exchange_group)")
    os.environ["NCCL_MAX_NCHANNELS"] = str(reserve)
    return reserve


_EXCHANGE_GROUPS = {}


def exchange_group(reserve=None):
    """the process group the gradient buckets travel on.  RCCL: a communicator of its OWN over all ranks whose
    ncclConfig_t caps its workgroups (`max_ctas`) at the CU reserve -- independent of the environment and of whatever
    communicator the launcher (Lightning, torchrun user code) created first; collective over all ranks, like
    `dist.new_group`.  Other backends, or a torch without the option: the default group (None)."""
    if not _initialized() or not backend_runs_on_gpu():
        return None
    reserve = reserve_setting() if reserve is None else int(reserve)
    if reserve <= 0 or os.environ.get("SGDM_EXCHANGE_GROUP", "1") == "0":
        return None
    key = (reserve, dist.get_world_size())
    if key in _EXCHANGE_GROUPS:
        return _EXCHANGE_GROUPS[key]
    grp = None
    try:
        opts = dist.ProcessGroupNCCL.Options()
        # RCCL accepts 1 .. 64 workgroups per communicator (its MAXCHANNELS); a larger reserve still caps at 64
        ctas = max(1, min(int(reserve), 64))
        opts.config.max_ctas = ctas
        opts.config.min_ctas = min(4, ctas)
        grp = dist.new_group(ranks=list(range(dist.get_world_size())), backend="nccl", pg_options=opts)
        # ProcessGroupNCCL creates the communicator lazily: force it NOW, inside the try and outside any timed or overlapped
        # region -- a configuration RCCL rejects then falls back to the default group here instead of failing at the first
        # bucket in the middle of the first backward, and ncclCommInit does not land inside that backward either
        probe = torch.zeros(1, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(probe, group=grp)
        torch.cuda.current_stream().synchronize()
    except Exception as exc:                          # pragma: no cover - depends on the torch / RCCL build
        warnings.warn(f"sgdm_amd: no per-communicator CTA cap available ({exc}); the exchange uses the default group")
        grp = None
    _EXCHANGE_GROUPS[key] = grp
    return grp
