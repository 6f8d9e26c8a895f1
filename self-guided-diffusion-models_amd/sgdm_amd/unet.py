"""Drop-in UNet denoisers (``dynamic=unet_fast`` / ``dynamic=unetca_fast``) on hand-written HIP kernels.

Host side of the operator boundary of SURVEY.md 8(b): same constructor keywords, method signatures,
``state_dict`` names/shapes/order and return values as

    dynamic.diffusionmodules.openaimodel.UNetModel      (reference openaimodel.py:466-956)
    dynamic.diffusionmodules.openaimodel_ca.UNetModel   (reference openaimodel_ca.py:449-1033)

but ``forward`` is a static program of launches into libsgdm_hip.so (include/sgdm_hip.h): NHWC fp32
activations, GroupNorm/SiLU/FiLM, resampling, skip-concat, bias and residual fused into the
implicit-GEMM convolution's loader/epilogue; attention cores as fused MFMA kernels.  PyTorch is used
for parameter storage, device memory and streams only.  There is no CPU / eager fallback: without the
HIP library or a GPU the forward raises.
"""
import ctypes as C
import weakref
import math
import os

import torch
import torch.nn as nn

from . import _lib as L

GN_GROUPS, GN_EPS, LN_EPS = 32, 1e-5, 1e-5
NUM_TIME_TOKENS = NUM_COND_TOKENS = 8


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timestep_freqs(dim, max_period=10000):
    """frequency table of timestep_embedding, same fp32 op order as the reference (util.py:161-164)"""
    half = dim // 2
    return torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)


def default_precision():
    return os.environ.get("SGDM_PREC", "f32")


# ------------------------------------------------------------------------------------------------
# parameter containers: a generic nn.Module tree so that state_dict() keys equal the reference's
# ------------------------------------------------------------------------------------------------
class _Node(nn.Module):
    pass


def _register(root, name, shape, kind, init):
    parts = name.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    t = init(shape)
    if kind == "buffer":
        node.register_buffer(parts[-1], t)
    else:
        node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=(kind == "param")))


def _get(root, name):
    obj = root
    for p in name.split("."):
        obj = getattr(obj, p) if not p.isdigit() else obj._modules[p]
    return obj


# initialisers following the reference modules' defaults
def _init_weight(shape):
    w = torch.empty(shape)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))          # nn.Linear / nn.Conv default
    return w


def _init_bias_for(fan_in):
    def f(shape):
        b = torch.empty(shape)
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        nn.init.uniform_(b, -bound, bound)
        return b
    return f


_zeros = lambda s: torch.zeros(s)
_ones = lambda s: torch.ones(s)
_randn = lambda s: torch.randn(s)


class _Spec:
    """ordered parameter list builder"""

    def __init__(self):
        self.items = []

    def linear(self, p, i, o, bias=True, zero=False):
        self.items.append((p + ".weight", (o, i), "param", _zeros if zero else _init_weight))
        if bias:
            self.items.append((p + ".bias", (o,), "param", _zeros if zero else _init_bias_for(i)))

    def conv(self, p, i, o, k, dims=2, zero=False):
        self.items.append((p + ".weight", (o, i) + (k,) * dims, "param", _zeros if zero else _init_weight))
        self.items.append((p + ".bias", (o,), "param", _zeros if zero else _init_bias_for(i * k ** dims)))

    def norm(self, p, c):
        self.items.append((p + ".weight", (c,), "param", _ones))
        self.items.append((p + ".bias", (c,), "param", _zeros))


# ------------------------------------------------------------------------------------------------
# packed weights
# ------------------------------------------------------------------------------------------------
AMAX_OF = {}       # parameter data_ptr -> (version, device int32 holding the bits of max |w|, weakref to the parameter:
                   # an address can be recycled by another tensor), written by _Packed.refresh


HEAD_DIMS = (16, 32, 64, 128)          # head widths the MFMA attention cores are instantiated for


def padded_head_dim(d):
    """smallest attention-core head width >= d: other widths (config/dynamic/unetca_fast_s64.yaml: 672 / 32 = 21 and
    896 / 32 = 28 channels per head, openaimodel_ca.py:671-693) run zero-padded -- zeros add nothing to q.k and carry
    no value -- with the softmax scale of the TRUE width"""
    for dp in HEAD_DIMS:
        if dp >= d:
            return dp
    raise ValueError(f"attention head width {d} > {HEAD_DIMS[-1]} is not supported")


class _Pad:
    """zero-padded re-layout of a 2-D weight (or a vector / [2, d] table along its last dim): source row r lands at row
    rows[r], source column c at column cols[c] of an [n_rows, n_cols] zero matrix.  ``gather`` is the adjoint (the
    gradient of the padded tensor read back at the positions of the real entries)."""

    def __init__(self, rows=None, n_rows=None, cols=None, n_cols=None):
        self.rows = None if rows is None else torch.as_tensor(rows, dtype=torch.int64)
        self.cols = None if cols is None else torch.as_tensor(cols, dtype=torch.int64)
        self.n_rows, self.n_cols = n_rows, n_cols

    def _idx(self, dev):
        if self.rows is not None and self.rows.device != dev:
            self.rows = self.rows.to(dev)
        if self.cols is not None and self.cols.device != dev:
            self.cols = self.cols.to(dev)

    def apply(self, w):
        self._idx(w.device)
        w2 = w.reshape(w.shape[0], -1)
        out = w2.new_zeros(self.n_rows if self.rows is not None else w2.shape[0],
                           self.n_cols if self.cols is not None else w2.shape[1])
        if self.rows is not None and self.cols is not None:
            out[self.rows[:, None], self.cols[None, :]] = w2
        elif self.rows is not None:
            out[self.rows] = w2
        else:
            out[:, self.cols] = w2
        return out

    def gather(self, wp):
        self._idx(wp.device)
        if self.rows is not None:
            wp = wp[self.rows]
        if self.cols is not None:
            wp = wp[:, self.cols]
        return wp


class _Packed:
    """device buffer holding one conv/linear weight in the igemm layout, refreshed when the source
    parameter(s) change (optimizer step / load_state_dict bump ``_version``)."""

    def __init__(self, srcs, ksize, prec, pad=None):
        self.srcs = srcs                     # list of parameters concatenated along dim 0
        self.ksize = ksize
        self.prec = prec
        self.pad = pad                       # _Pad: the packed operator is the zero-padded re-layout of the parameter
        self.cout = sum(s.shape[0] for s in srcs)
        self.cin = srcs[0].shape[1]
        if pad is not None:
            assert ksize == 1
            self.cout = pad.n_rows if pad.rows is not None else self.cout
            self.cin = pad.n_cols if pad.cols is not None else self.cin
        lib = L.load()
        nbytes = lib.sgd_packed_weight_bytes(self.cout, self.cin, ksize, prec)
        self.buf = torch.empty(nbytes // 4, dtype=torch.float32, device=srcs[0].device)
        self.cin_p = self.cout_p = 0
        self.sig = None
        # split precisions: per-tensor power-of-two scale so hi AND lo halves stay in fp16's normal range whatever the
        # tensor's magnitude (include/sgdm_hip.h: sgd_igemm_args.w_scale_inv); formed on the device, no host round trip
        self.scaled = prec != L.PREC_F32
        if self.scaled:
            self.amax = torch.zeros(1, dtype=torch.int32, device=srcs[0].device)
            self.scale_inv = torch.ones(1, dtype=torch.float32, device=srcs[0].device)

    @property
    def scale_ptr(self):
        return self.scale_inv.data_ptr() if self.scaled else 0

    def refresh(self, stream):
        sig = tuple((s.data_ptr(), s._version) for s in self.srcs)
        if sig == self.sig:
            return
        lib = L.load()
        src = self.srcs[0].detach() if len(self.srcs) == 1 else torch.cat([s.detach() for s in self.srcs], 0)
        src = src.contiguous().float()
        if self.pad is not None:
            src = self.pad.apply(src).contiguous()
        cin_p, cout_p = C.c_int32(0), C.c_int32(0)
        if self.scaled:
            self.amax.zero_()
            L.check(lib.sgd_weight_amax(_ptr(src), src.numel(), _ptr(self.amax), stream), "sgd_weight_amax")
            if len(self.srcs) == 1 and self.pad is None:
                # max |w| of this parameter at this version: the adjoint pack of the same tensor (train._PackedAdj) reuses it
                AMAX_OF[self.srcs[0].data_ptr()] = (self.srcs[0]._version, self.amax, weakref.ref(self.srcs[0]))
            L.check(lib.sgd_pack_weight_scaled(_ptr(src), _ptr(self.buf), self.cout, self.cin, self.ksize, self.prec, 0,
                                               _ptr(self.amax), _ptr(self.scale_inv), C.byref(cin_p), C.byref(cout_p),
                                               stream), "sgd_pack_weight_scaled")
        else:
            L.check(lib.sgd_pack_weight(_ptr(src), _ptr(self.buf), self.cout, self.cin, self.ksize, self.prec,
                                        C.byref(cin_p), C.byref(cout_p), stream), "sgd_pack_weight")
        self.cin_p, self.cout_p = cin_p.value, cout_p.value
        self._keep = src
        self.sig = sig


class _PackBatch:
    """the weight packs of one launch program refreshed together: when (almost) all of them are stale -- every step of a
    training loop -- ONE sgd_pack_weights_batched call (zero + amax + pack kernels over a device-side job table) replaces
    two launches per tensor (142 pack + 74 amax launches of ~6 us per step at C2).  Members are `_Packed` /
    `train._PackedAdj` objects whose source is one whole contiguous fp32 parameter; the others (concatenated FiLM
    projections, zero-padded head re-layouts, sliced sources) keep their own refresh."""

    MIN_STALE = 8           # fewer stale packs than this: refresh them one by one

    def __init__(self, packs, prec, dev):
        self.packs, self.prec, self.dev = list(packs), prec, dev
        self.lib = L.load()
        self.members, self.table_sig = None, None

    @staticmethod
    def _source(pk):
        """(parameter, forward cout, forward cin, transpose) of a batchable pack, else None"""
        if hasattr(pk, "srcs"):                                   # unet._Packed
            if len(pk.srcs) != 1 or pk.pad is not None:
                return None
            p, t = pk.srcs[0], 0
            cout, cin = pk.cout, pk.cin
        else:                                                     # train._PackedAdj
            if len(pk.deps) != 1:
                return None
            p, t = pk.deps[0], 1
            src = pk.src_fn()
            if src.data_ptr() != p.data_ptr() or src.numel() != p.numel():
                return None
            cout, cin = pk.cout_fwd, pk.cin_fwd
        if p.dtype != torch.float32 or not p.is_contiguous() or p.numel() != cout * cin * pk.ksize * pk.ksize:
            return None
        return p, cout, cin, t

    def _build(self):
        members, jobs = [], []
        ab, pb = C.c_int32(), C.c_int32()
        cp, op = C.c_int32(), C.c_int32()
        a_job, a_first, p_job, p_first = [], [0], [], [0]
        scaled = self.prec != L.PREC_F32
        for pk in self.packs:
            srcd = self._source(pk)
            if srcd is None:
                continue
            p, cout, cin, t = srcd
            L.check(self.lib.sgd_pack_job_blocks(cout, cin, pk.ksize, self.prec, t, C.byref(ab), C.byref(pb), C.byref(cp),
                                                 C.byref(op)), "sgd_pack_job_blocks")
            j = len(members)
            jobs.append(L.PackJob(src=p.data_ptr(), dst=pk.buf.data_ptr(), amax_bits=pk.amax.data_ptr() if scaled else 0,
                                  scale_inv=pk.scale_inv.data_ptr() if scaled else 0, cout=cout, cin=cin, ksize=pk.ksize,
                                  transpose=t, own_amax=1 if scaled else 0))
            members.append((pk, p, cp.value, op.value))
            a_job += [j] * (ab.value if scaled else 0)
            a_first.append(len(a_job))
            p_job += [j] * pb.value
            p_first.append(len(p_job))
        self.members = members
        self.table_sig = tuple(p.data_ptr() for _, p, _, _ in members)
        if not members:
            return
        arr = (L.PackJob * len(jobs))(*jobs)
        self.jobs = torch.frombuffer(bytearray(C.string_at(C.addressof(arr), C.sizeof(arr))), dtype=torch.uint8).to(self.dev)
        i32t = lambda v: torch.tensor(v if v else [0], dtype=torch.int32, device=self.dev)
        self.a_job, self.a_first, self.p_job, self.p_first = i32t(a_job), i32t(a_first), i32t(p_job), i32t(p_first)
        self.n_a, self.n_p = len(a_job), len(p_job)
        self.batched = {id(pk) for pk, _, _, _ in members}

    def refresh(self, stream):
        if self.members is None or self.table_sig != tuple(p.data_ptr() for _, p, _, _ in self.members):
            self._build()                                          # first use, or a parameter moved (.to(), re-materialised)
        stale = [(pk, p, cp, op) for pk, p, cp, op in self.members if pk.sig != ((p.data_ptr(), p._version),)]
        if len(stale) >= self.MIN_STALE and os.environ.get("SGDM_BATCHED_PACK", "1") != "0":
            L.check(self.lib.sgd_pack_weights_batched(_ptr(self.jobs), len(self.members), _ptr(self.a_job), _ptr(self.a_first),
                                                      self.n_a, _ptr(self.p_job), _ptr(self.p_first), self.n_p, self.prec,
                                                      stream), "sgd_pack_weights_batched")
            for pk, p, cp, op in self.members:                     # (fresh members were re-packed too: same bytes)
                pk.cin_p, pk.cout_p = cp, op
                pk.sig = ((p.data_ptr(), p._version),)
                if hasattr(pk, "srcs") and pk.scaled:
                    AMAX_OF[p.data_ptr()] = (p._version, pk.amax, weakref.ref(p))
        for pk in self.packs:                                      # the rest, and small stale sets, one by one
            pk.refresh(stream)


class _Program:
    def __init__(self):
        self.ops = []
        self.meta = []
        self.keep = []

    def add(self, name, fn, *args, flops=0.0, nbytes=0.0):
        self.ops.append((name, fn, args))
        self.meta.append((getattr(fn, "__name__", "op"), flops, nbytes))

    def run(self, stream):
        for name, fn, args in self.ops:
            rc = fn(*args, stream)
            if rc:
                L.check(rc, name)

    def run_profiled(self, stream):
        """same launches with a HIP event pair around each (events on the launch stream).
        Returns [(tag, symbol, ms, algorithmic flops, algorithmic bytes)]."""
        evs = []
        for name, fn, args in self.ops:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args, stream)
            e1.record()
            if rc:
                L.check(rc, name)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        return [(op[0], mt[0], e0.elapsed_time(e1), mt[1], mt[2])
                for op, mt, (e0, e1) in zip(self.ops, self.meta, evs)]


# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------
class UNetModelBase(nn.Module):
    KIND = None

    # -------------------------------------------------------------- construction (reference ctor logic)
    def _setup(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
               attention_resolutions, dropout, channel_mult, conv_resample, dims, use_checkpoint, use_fp16,
               num_heads, num_head_channels, num_heads_upsample, use_scale_shift_norm, resblock_updown,
               cond_dim, condition, condition_method):
        if dims != 2:
            raise NotImplementedError("HIP path implements dims=2 only")
        if use_fp16:
            # (the reference's own forward raises with the flag set -- `h = x.type(torch.float16)` meets fp32 weights,
            # openaimodel.py:564,926: "Input type (c10::Half) and bias type (float) should be the same" -- unless a caller has run
            # convert_to_fp16() (:837), and nothing in the reference calls it)
            raise NotImplementedError("use_fp16=True is not supported (reference configs use fp32)")
        if num_heads == -1 and num_head_channels == -1:
            raise AssertionError("Either num_heads or num_head_channels has to be set")
        if not conv_resample:
            raise NotImplementedError("HIP path implements conv_resample=True (ctor default)")
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        self.dropout = dropout
        self.channel_mult = tuple(channel_mult)
        self.conv_resample = conv_resample
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float32
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads if num_heads_upsample == -1 else num_heads_upsample
        self.use_scale_shift_norm = use_scale_shift_norm
        self._film_mult = 2 if use_scale_shift_norm else 1        # emb_layers' width per ResBlock channel (openaimodel.py:259-264)
        self.resblock_updown = resblock_updown
        self.cond_dim = 0 if cond_dim is None else cond_dim
        self.condition = condition
        self.condition_method = condition_method
        self.hip_precision = default_precision()
        self._engines = {}

    def _heads_for(self, ch, up=False):
        """input/middle blocks use num_heads, output blocks num_heads_upsample (openaimodel.py:664-668,778-782)"""
        if self.num_head_channels == -1:
            return self.num_heads_upsample if up else self.num_heads
        return ch // self.num_head_channels

    def _walk(self):
        """block plan; mirrors openaimodel.py:634-835 / openaimodel_ca.py:645-836"""
        mc, cm, nrb = self.model_channels, self.channel_mult, self.num_res_blocks
        inp = [[("conv", self._in_ch_total, mc)]]
        chans = [mc]
        ch, ds = mc, 1
        for level, mult in enumerate(cm):
            for _ in range(nrb):
                layers = [("res", ch, mult * mc, None)]
                ch = mult * mc
                if ds in self.attention_resolutions:
                    layers.append(("attn", ch, self._heads_for(ch)))
                inp.append(layers)
                chans.append(ch)
            if level != len(cm) - 1:
                inp.append([("res", ch, ch, "down")] if self.resblock_updown else [("down", ch)])
                chans.append(ch)
                ds *= 2
        mid = [("res", ch, ch, None), ("attn", ch, self._heads_for(ch)), ("res", ch, ch, None)]
        out = []
        for level, mult in list(enumerate(cm))[::-1]:
            for i in range(nrb + 1):
                ich = chans.pop()
                layers = [("res", ch + ich, mc * mult, None)]
                ch = mc * mult
                if ds in self.attention_resolutions:
                    layers.append(("attn", ch, self._heads_for(ch, up=True)))
                if level and i == nrb:
                    layers.append(("res", ch, ch, "up") if self.resblock_updown else ("up", ch))
                    ds //= 2
                out.append(layers)
        return inp, mid, out

    def _layer_spec(self, sp, p, layer):
        kind = layer[0]
        if kind == "conv":
            sp.conv(p, layer[1], layer[2], 3)
        elif kind == "res":
            _, cin, cout, _ud = layer
            sp.norm(p + ".in_layers.0", cin)
            sp.conv(p + ".in_layers.2", cin, cout, 3)
            sp.linear(p + ".emb_layers.1", self._emb_ch, self._film_mult * cout)      # openaimodel.py:259-264
            sp.norm(p + ".out_layers.0", cout)
            sp.conv(p + ".out_layers.3", cout, cout, 3, zero=True)        # zero_module, openaimodel.py:273-276
            if cin != cout:
                sp.conv(p + ".skip_connection", cin, cout, 1)
        elif kind == "attn":
            self._attn_spec(sp, p, layer[1], layer[2])
        elif kind == "down":
            sp.conv(p + ".op", layer[1], layer[1], 3)
        elif kind == "up":
            sp.conv(p + ".conv", layer[1], layer[1], 3)

    def _register_all(self, sp_head):
        sp = sp_head
        inp, mid, out = self._walk()
        self._plan = (inp, mid, out)
        for i, blk in enumerate(inp):
            for j, layer in enumerate(blk):
                self._layer_spec(sp, f"input_blocks.{i}.{j}", layer)
        for j, layer in enumerate(mid):
            self._layer_spec(sp, f"middle_block.{j}", layer)
        for i, blk in enumerate(out):
            for j, layer in enumerate(blk):
                self._layer_spec(sp, f"output_blocks.{i}.{j}", layer)
        sp.norm("out.0", self.model_channels)
        sp.conv("out.2", self.model_channels, self.out_channels, 3, zero=True)   # openaimodel.py:833-834
        for name, shape, kind, init in sp.items:
            _register(self, name, shape, kind, init)
        self._manifest = [(n, tuple(s), k) for n, s, k, _ in sp.items]

    def P(self, name):
        return _get(self, name)

    # -------------------------------------------------------------- reference API
    def get_guided_score(self, z, zc, w):
        st = self.condition["scale_type"] if isinstance(self.condition, dict) else self.condition.scale_type
        if st == "imagen":
            return (1 - w) * z + w * zc
        elif st == "cfg":
            return (1 + w) * zc - w * z
        raise ValueError(st)

    def _scale_mode(self):
        st = self.condition["scale_type"] if isinstance(self.condition, dict) else self.condition.scale_type
        if st == "imagen":
            return 1
        if st == "cfg":
            return 2
        raise ValueError(st)

    def convert_to_fp16(self):
        pass

    def convert_to_fp32(self):
        pass

    # -------------------------------------------------------------- engine
    def _engine(self, n, h, w, prec):
        dev = self._device()
        sig = tuple(p.data_ptr() for p in self._all_tensors())
        key = (n, h, w, prec)
        eng = self._engines.get(key)
        if eng is None or eng.sig != sig:
            eng = _Engine(self, n, h, w, prec, dev)
            eng.sig = sig
            self._engines[key] = eng
        return eng

    def _all_tensors(self):
        if not hasattr(self, "_tensor_list") or self._tensor_list_n != len(self._manifest):
            self._tensor_list = [self.P(n) for n, _, _ in self._manifest]
            self._tensor_list_n = len(self._manifest)
        return self._tensor_list

    def _device(self):
        return self.P(self._manifest[-1][0]).device

    def _check_runnable(self, x):
        if x.device.type != "cuda":
            raise RuntimeError("sgdm_amd UNet runs on the MI355X HIP path only (inputs must be on a cuda "
                               "device); there is no CPU fallback")
        L.load()

    def _draw_mask(self, n, cond_drop_prob, device):
        """prob_mask_like (openaimodel.py:462-463): same RNG consumption as the reference"""
        return torch.zeros((n,), device=device).float().uniform_(0, 1) < cond_drop_prob

    def _run(self, x, t, cond, layout, mask, n):
        """one UNet evaluation at UNet batch n (inputs have n_src = len(x) rows, n % n_src == 0).
        Returns the engine (eps in eng.eps_nhwc [n, H, W, out_channels])."""
        self._check_runnable(x)
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            from .train import forward_train      # autograd-capable path
            return forward_train(self, x, t, cond, layout, mask, n)
        B, cx, H, W = x.shape
        assert cx == self.in_channels
        prec = L.PREC_BY_NAME[self.hip_precision]
        eng = self._engine(n, H, W, prec)
        eng.run(x, t, cond, layout, mask)
        return eng

    def _to_nchw(self, eng):
        lib = L.load()
        n, H, W = eng.n, eng.h, eng.w
        out = torch.empty(n, self.out_channels, H, W, device=eng.dev, dtype=torch.float32)
        L.check(lib.sgd_nhwc_to_nchw(_ptr(eng.eps_nhwc), n, H, W, self.out_channels, _ptr(out),
                                     torch.cuda.current_stream().cuda_stream), "sgd_nhwc_to_nchw")
        return out

    def _cfg_combine(self, eng, w, B):
        lib = L.load()
        out = torch.empty(B, self.out_channels, eng.h, eng.w, device=eng.dev, dtype=torch.float32)
        L.check(lib.sgd_cfg_combine(_ptr(eng.eps_nhwc), self._scale_mode(), float(w), B, self.out_channels,
                                    eng.h * eng.w, _ptr(out), torch.cuda.current_stream().cuda_stream),
                "sgd_cfg_combine")
        return out


def device_cus(dev):
    """compute units of `dev` (256 on MI355X)"""
    return int(torch.cuda.get_device_properties(dev).multi_processor_count)


# ------------------------------------------------------------------------------------------------
_GRAD_HEALTH = {}


def grad_health(dev):
    """device float[1], one per device and process: != 0 while a training backward on this device has produced invalid
    gradients that the host has not yet acknowledged (balanced-tail time-out, summed over the ranks of a data-parallel
    job).  train.Backward.run adds to it, sgd_adamw_ema_step takes it as its `skip_if_nonzero` (optim.FusedAdamWEma), the
    engine's poll_health() raises and clears it."""
    dev = torch.device(dev)
    key = (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))
    if key not in _GRAD_HEALTH:
        _GRAD_HEALTH[key] = torch.zeros(1, dtype=torch.float32, device=dev)
    return _GRAD_HEALTH[key]


class _Engine:
    """static launch program + workspace for one (UNet batch, H, W, precision)"""

    def __init__(self, model, n, h, w, prec, dev):
        self.m, self.n, self.h, self.w, self.prec, self.dev = model, n, h, w, prec, dev
        self.lib = L.load()
        self.prog = _Program()
        self.packed = []
        self.bufs = []
        self.tape = []            # forward structure, walked in reverse by train.build_backward
        self.drops = []           # (igemm args, p box, seed box) of the launches that apply train-time dropout
        self.lse = {}             # attention log-sum-exp buffers (written only when present)
        self.stats_of = {}        # tensor data_ptr -> (partial statistics buffer, parts, channels) written by its producer
        self._p_drop = 0.0
        self._grid_cap = 0        # persistent-grid cap of the backward program's launches (set_grid_cap)
        self.refresh_hooks = []   # derived device state that follows the parameters (run by refresh(), never captured)
        # scratch of sgd_igemm's balanced tail (partial accumulators of K-split tiles + self-resetting arrival counters):
        # zeroed once, shared by every launch of this engine's programs -- they are ordered on one stream
        self.work_bytes = int(self.lib.sgd_igemm_work_bytes()) if os.environ.get("SGDM_BALANCE", "1") != "0" else 0
        self.work = torch.zeros(max(1, self.work_bytes // 4), dtype=torch.float32, device=dev)
        self._build()

    # ---- helpers
    def buf(self, *shape, dtype=torch.float32):
        t = torch.empty(*shape, dtype=dtype, device=self.dev)
        self.bufs.append(t)
        return t

    def attention_fn(self):
        """the attention core in the engine's arithmetic: split-precision MFMA for f16x3, exact fp32 otherwise
        (bf16x3 keeps the exact kernel: 8 mantissa bits per half do not hold the softmax weights)"""
        return self.lib.sgd_attention_split if self.prec == L.PREC_F16X3 else self.lib.sgd_attention

    def pack(self, names, ksize, pad=None):
        pk = _Packed([self.m.P(nm) for nm in names], ksize, self.prec, pad)
        self.packed.append(pk)
        return pk

    def padded(self, name, pad):
        """device copy of parameter `name` in a zero-padded layout (_Pad over its last dim for vectors / tables), kept in
        step with the parameter by refresh() like the packed weights"""
        src = self.m.P(name)
        buf = self.buf(*pad.apply(src.detach().reshape(-1, src.shape[-1]) if src.dim() > 1 else src.detach().reshape(1, -1)).shape)
        box = dict(sig=None)

        def hook(stream):
            sig = (src.data_ptr(), src._version)
            if sig != box["sig"]:
                v = src.detach().float()
                buf.copy_(pad.apply(v.reshape(-1, v.shape[-1]) if v.dim() > 1 else v.reshape(1, -1)))
                box["sig"] = sig
        self.refresh_hooks.append(hook)
        return buf

    def igemm(self, tag, x0, c0, y, cout, pk, *, x1=None, c1=0, conv=None, m=0, rows_per_n=0,
              pro=L.PRO_NONE, silu=0, pa=None, pb=None, pc=None, bias=None, res=None, res_mode=L.RS_NONE,
              y_ld=None, y_off=0, orows=(0, 0, 0), stats=False, launch=True):
        """stats=True: the epilogue also writes the GroupNorm statistics of y (consumed by gn() through
        sgd_stats_reduce instead of a sgd_chan_stats pass over the tensor)"""
        a = L.IgemmArgs()
        a.x0, a.x1, a.c0, a.c1 = x0.data_ptr(), (x1.data_ptr() if x1 is not None else 0), c0, c1
        rows_n = conv[0] if conv is not None else (m // rows_per_n if rows_per_n else 0)
        if conv is not None:
            nimg, hi, wi, ho, wo, stride, resample = conv
            a.mode, a.n, a.hi, a.wi, a.ho, a.wo, a.stride, a.resample = L.MODE_CONV3, nimg, hi, wi, ho, wo, stride, resample
        else:
            a.mode, a.m, a.rows_per_n, a.stride = L.MODE_FLAT, m, rows_per_n, 1
        a.pro, a.pro_silu = pro, silu
        a.pa = pa.data_ptr() if pa is not None else 0
        a.pb = pb.data_ptr() if pb is not None else 0
        a.pc = pc.data_ptr() if pc is not None else 0
        a.w = pk.buf.data_ptr()
        a.w_scale_inv = pk.scale_ptr
        a.bias = bias.data_ptr() if bias is not None else 0
        a.res = res.data_ptr() if res is not None else 0
        a.res_mode = res_mode
        a.y = y.data_ptr() + 4 * y_off
        a.cout = cout
        a.y_ld = y_ld if y_ld is not None else cout
        a.orows_in, a.orows_out, a.orow_off = orows
        a.prec = self.prec
        if self.work_bytes:
            a.work, a.work_bytes = self.work.data_ptr(), self.work_bytes
        # the stem (3 / 4 input channels, no prologue, no residual): a plain fp32 kernel of its own (csrc/narrow.hip) reading
        # the PARAMETER -- on the implicit-GEMM kernel 29 of every 32 K lanes of this layer multiply zeros.  The descriptor
        # `a` is still built (and the weight still packed): the backward's weight gradient reads both.
        narrow = (conv is not None and launch and c1 == 0 and c0 in (3, 4) and pro == L.PRO_NONE and not silu and res is None
                  and conv[5] == 1 and conv[6] == 0 and cout % 4 == 0 and cout <= 1024 and y_off == 0 and orows == (0, 0, 0)
                  and len(pk.srcs) == 1 and pk.pad is None and pk.srcs[0].dtype == torch.float32 and pk.srcs[0].is_contiguous()
                  and os.environ.get("SGDM_NARROW_CONV", "1") != "0")
        # the head (3 / 4 OUTPUT channels behind a GroupNorm + SiLU): a plain fp32 kernel of its own too (csrc/narrow.hip,
        # round 5) -- on the implicit-GEMM kernel its 3 columns ride a 32-column tile.  Its weight operand is a [tap][co][ci]
        # copy of the parameter, kept in step by refresh()
        narrow_out = (conv is not None and launch and not narrow and c1 == 0 and cout in (3, 4) and c0 % 32 == 0 and res is None
                      and pro in (L.PRO_NONE, L.PRO_AFFINE_NC) and conv[5] == 1 and conv[6] == 0 and y_off == 0
                      and orows == (0, 0, 0) and not stats and len(pk.srcs) == 1 and pk.pad is None
                      and pk.srcs[0].dtype == torch.float32 and os.environ.get("SGDM_NARROW_CONV", "1") != "0")
        if stats and os.environ.get("SGDM_FUSED_STATS", "1") != "0":
            parts = self.lib.sgd_conv3_narrow_in_parts(conv[3], conv[4]) if narrow else self.lib.sgd_igemm_stats_parts(C.byref(a))
            if parts > 0:
                sbuf = self.buf(rows_n, parts, 2, cout)
                a.stats = sbuf.data_ptr()
                self.stats_of[y.data_ptr()] = (sbuf, parts, cout)
        self.prog.keep.append((a, pk))
        self._late.append((a, pk))                # cin_p / cout_p are known after the first pack
        # algorithmic work of this launch: 2*M*N*K flops; every input/output element moved once + weights
        rows = conv[0] * conv[3] * conv[4] if conv is not None else m
        rows_in = conv[0] * conv[1] * conv[2] if conv is not None else m
        taps = 9 if conv is not None else 1
        cin = c0 + c1
        flops = 2.0 * rows * cout * taps * cin
        nbytes = 4.0 * (rows_in * cin + rows * cout * (2 if res is not None else 1) + taps * cin * cout)
        if narrow:
            wsrc, fn = pk.srcs[0], self.lib.sgd_conv3_narrow_in
            x_p, b_p, y_p, s_p = _ptr(x0), C.c_void_p(a.bias or 0), C.c_void_p(a.y), C.c_void_p(a.stats or 0)
            nimg, ho, wo, y_ld = conv[0], conv[3], conv[4], a.y_ld

            def sgd_conv3_narrow_in(stream):          # the parameter's address is read at launch time, like a repack would
                return fn(x_p, C.c_void_p(wsrc.data_ptr()), b_p, y_p, s_p, nimg, ho, wo, c0, cout, y_ld, 0, stream)
            self.prog.add(tag, sgd_conv3_narrow_in, flops=flops, nbytes=nbytes)
        elif narrow_out:
            wsrc, fn = pk.srcs[0], self.lib.sgd_conv3_narrow_out
            w9 = self.buf(9, cout, c0)
            box = dict(sig=None)

            def refresh_w9(stream, w9=w9, wsrc=wsrc, box=box):
                sig = (wsrc.data_ptr(), wsrc._version)
                if sig != box["sig"]:
                    w9.copy_(wsrc.detach().float().permute(2, 3, 0, 1).reshape(9, cout, c0))
                    box["sig"] = sig
            self.refresh_hooks.append(refresh_w9)
            x_p, y_p, b_p, w_p = _ptr(x0), C.c_void_p(a.y), C.c_void_p(a.bias or 0), _ptr(w9)
            pa_p, pb_p = C.c_void_p(a.pa or 0), C.c_void_p(a.pb or 0)
            nimg, ho, wo, y_ld = conv[0], conv[3], conv[4], a.y_ld

            def sgd_conv3_narrow_out(stream):
                return fn(x_p, pa_p, pb_p, int(silu), w_p, b_p, y_p, nimg, ho, wo, c0, cout, y_ld, stream)
            self.prog.add(tag, sgd_conv3_narrow_out, flops=flops, nbytes=nbytes)
        elif launch:        # launch=False: descriptor only (the backward's weight gradient reads it)
            self.prog.add(tag, self.lib.sgd_igemm, C.byref(a), flops=flops, nbytes=nbytes)
        return a

    def gn(self, tag, srcs, hw, gname, film=None, film_ld=0, eps=GN_EPS):
        """srcs: list of (tensor, channels) forming a virtual concat -> (a, b) coefficient buffers"""
        n = self.n
        ct = sum(c for _, c in srcs)
        sums = self.buf(n, ct, 2)
        off = 0
        parts = []
        for t, c in srcs:
            st = self.stats_of.get(t.data_ptr())
            if st is not None and st[2] == c:
                parts.append((st[0], st[1], c))           # partial statistics written by the producer's epilogue
            else:
                self.prog.add(tag + ".stats", self.lib.sgd_chan_stats, _ptr(t), n, hw, c, _ptr(sums), ct, off)
                parts.append((None, 0, c))
            off += c
        a, b = self.buf(n, ct), self.buf(n, ct)
        self._last_sums = sums
        gw, gb = self.m.P(gname + ".weight"), self.m.P(gname + ".bias")
        if len(parts) <= 2 and any(p[1] > 0 for p in parts):
            # fold the partials and form the coefficients in one launch
            (p0, n0, c0), (p1, n1, c1) = parts[0], (parts[1] if len(parts) == 2 else (None, 0, 0))
            self.prog.add(tag + ".coef", self.lib.sgd_gn_coef_parts, _ptr(p0) if p0 is not None else None, n0, c0,
                          _ptr(p1) if p1 is not None else None, n1, c1, _ptr(sums), _ptr(gw), _ptr(gb),
                          C.c_void_p(film or 0), film_ld, n, GN_GROUPS, hw, eps, _ptr(a), _ptr(b))
        else:
            self.prog.add(tag + ".coef", self.lib.sgd_gn_coef, _ptr(sums), _ptr(gw), _ptr(gb), C.c_void_p(film or 0),
                          film_ld, n, ct, GN_GROUPS, hw, eps, _ptr(a), _ptr(b))
        return a, b

    # ---- program construction
    def _build(self):
        m, n, H, W = self.m, self.n, self.h, self.w
        self._late = []
        mc = m.model_channels
        ted = 4 * mc
        P = m.P
        lib = self.lib
        # ---------------- boundary buffers (filled by run())
        self.temb = self.buf(n, mc)
        self.freqs = timestep_freqs(mc).to(self.dev)
        cin_tot = m._in_ch_total
        self.x_in = self.buf(n, H, W, cin_tot)
        self.cond_m = self.buf(n, max(1, m._cond_width)) if m._cond_width else None
        # ---------------- embedding path
        # emb = [time part | cond part] is a VIRTUAL concat for unet_fast (openaimodel.py:942): two buffers
        emb = self.buf(n, ted)
        e1 = self.buf(n, ted)
        a0 = self.igemm("time_embed.0", self.temb, mc, e1, ted, self.pack(["time_embed.0.weight"], 1), m=n,
                        bias=P("time_embed.0.bias"))
        a2 = self.igemm("time_embed.2", e1, ted, emb, ted, self.pack(["time_embed.2.weight"], 1), m=n, silu=1,
                        bias=P("time_embed.2.bias"))
        self.tape.append(dict(kind="mlp2", name="time_embed", x=self.temb, h=e1, y=emb, a0=a0, a2=a2, cin=mc,
                              mid=ted, cout=ted, add_to_y=False))
        self.context = None
        self.emb_c = None
        m._build_cond_path(self, emb)
        # all ResBlocks' emb_layers in one GEMM (they share SiLU(emb))
        res_names = m._res_prefixes()
        fm = m._film_mult                     # 2: (scale, shift) of the FiLM form; 1: the additive form (use_scale_shift_norm=False)
        film_w = sum(fm * co for _, co in res_names)
        film = self.buf(n, film_w)
        fbias = self.buf(film_w)
        self._film_bias_src = [P(p + ".emb_layers.1.bias") for p, _ in res_names]
        self._film_bias = fbias
        ec = self.emb_c
        af = self.igemm("emb_layers", emb, ted, film, film_w,
                        self.pack([p + ".emb_layers.1.weight" for p, _ in res_names], 1), m=n, silu=1, bias=fbias,
                        x1=ec, c1=(m._emb_ch - ted) if ec is not None else 0)
        self.tape.append(dict(kind="film", emb_t=emb, emb_c=ec, film=film, film_w=film_w, a=af,
                              names=[p for p, _ in res_names], couts=[co for _, co in res_names]))
        self.film, self.film_ld = film, film_w
        self.film_off = {}
        off = 0
        for p, co in res_names:
            self.film_off[p] = off
            off += fm * co
        # ---------------- trunk
        inp, mid, out = m._plan
        hs = []
        cur = (self.x_in, cin_tot, H, W)
        for i, blk in enumerate(inp):
            cur = self._block(f"input_blocks.{i}", blk, [cur])
            hs.append(cur)
        cur = self._block("middle_block", mid, [cur])
        for i, blk in enumerate(out):
            cur = self._block(f"output_blocks.{i}", blk, [cur, hs.pop()])
        t, c, hh, ww = cur
        a, b = self.gn("out.0", [(t, c)], hh * ww, "out.0")
        self.eps_nhwc = self.buf(n, hh, ww, m.out_channels)
        ah = self.igemm("out.2", t, c, self.eps_nhwc, m.out_channels, self.pack(["out.2.weight"], 3),
                        conv=(n, hh, ww, hh, ww, 1, L.RS_NONE), pro=L.PRO_AFFINE_NC, silu=1, pa=a, pb=b,
                        bias=P("out.2.bias"))
        self.tape.append(dict(kind="head", x=t, c=c, hw=(hh, ww), a=a, b=b, sums=self._last_sums, conv=ah))

    def _block(self, prefix, blk, srcs):
        """srcs: list of (tensor, C, H, W) (two entries = virtual concat [h, skip])"""
        for j, layer in enumerate(blk):
            p = f"{prefix}.{j}"
            kind = layer[0]
            if kind == "conv":
                (t, c, hh, ww), = srcs
                y = self.buf(self.n, hh, ww, layer[2])
                a = self.igemm(p, t, c, y, layer[2], self.pack([p + ".weight"], 3),
                               conv=(self.n, hh, ww, hh, ww, 1, L.RS_NONE), bias=self.m.P(p + ".bias"), stats=True)
                self.tape.append(dict(kind="conv_in", p=p, x=t, y=y, a=a, cin=c, cout=layer[2], hw=(hh, ww)))
                srcs = [(y, layer[2], hh, ww)]
            elif kind == "res":
                srcs = [self._res(p, layer, srcs)]
            elif kind == "attn":
                srcs = [self.m._build_attn(self, p, layer, srcs[0])]
            elif kind == "down":
                (t, c, hh, ww), = srcs
                y = self.buf(self.n, hh // 2, ww // 2, c)
                a = self.igemm(p + ".op", t, c, y, c, self.pack([p + ".op.weight"], 3),
                               conv=(self.n, hh, ww, hh // 2, ww // 2, 2, L.RS_NONE), bias=self.m.P(p + ".op.bias"),
                               stats=True)
                self.tape.append(dict(kind="down", p=p, x=t, y=y, c=c, hw_in=(hh, ww), a=a))
                srcs = [(y, c, hh // 2, ww // 2)]
            elif kind == "up":
                (t, c, hh, ww), = srcs
                y = self.buf(self.n, hh * 2, ww * 2, c)
                a = self.igemm(p + ".conv", t, c, y, c, self.pack([p + ".conv.weight"], 3),
                               conv=(self.n, hh, ww, hh * 2, ww * 2, 1, L.RS_UP2), bias=self.m.P(p + ".conv.bias"),
                               stats=True)
                self.tape.append(dict(kind="up", p=p, x=t, y=y, c=c, hw_in=(hh, ww), a=a))
                srcs = [(y, c, hh * 2, ww * 2)]
        return srcs[0]

    def _res(self, p, layer, srcs):
        """ResBlock._forward (openaimodel.py:300-320).  use_scale_shift_norm=True (every shipped plan): the embedding's (scale, shift)
        folded into out_layers' GroupNorm coefficients.  False (:317-319, h = h + emb_out): the embedding row added to conv1's output
        in place, the GroupNorm statistics by a pass of their own (conv1's epilogue statistics are those of h without it)."""
        _, cin, cout, ud = layer
        n, P = self.n, self.m.P
        t0, c0, hh, ww = srcs[0]
        t1, c1 = (srcs[1][0], srcs[1][1]) if len(srcs) == 2 else (None, 0)
        assert c0 + c1 == cin
        rs = {None: L.RS_NONE, "down": L.RS_AVGPOOL2, "up": L.RS_UP2}[ud]
        ho, wo = (hh // 2, ww // 2) if ud == "down" else ((hh * 2, ww * 2) if ud == "up" else (hh, ww))
        ss = self.m.use_scale_shift_norm
        a1, b1 = self.gn(p + ".in_layers.0", [(t0, c0)] + ([(t1, c1)] if t1 is not None else []), hh * ww,
                         p + ".in_layers.0")
        sums1 = self._last_sums
        h1 = self.buf(n, ho, wo, cout)
        ac1 = self.igemm(p + ".in_layers.2", t0, c0, h1, cout, self.pack([p + ".in_layers.2.weight"], 3), x1=t1,
                         c1=c1, conv=(n, hh, ww, ho, wo, 1, rs), pro=L.PRO_AFFINE_NC, silu=1, pa=a1, pb=b1,
                         bias=P(p + ".in_layers.2.bias"), stats=ss)
        film_ptr = self.film.data_ptr() + 4 * self.film_off[p]
        if ss:
            a2, b2 = self.gn(p + ".out_layers.0", [(h1, cout)], ho * wo, p + ".out_layers.0", film=film_ptr,
                             film_ld=self.film_ld)
        else:
            self.prog.add(p + ".emb_add", self.lib.sgd_add_rows_nc, _ptr(h1), C.c_void_p(film_ptr), self.film_ld, n, ho * wo, cout)
            a2, b2 = self.gn(p + ".out_layers.0", [(h1, cout)], ho * wo, p + ".out_layers.0")
        sums2 = self._last_sums
        ask = None
        if cin != cout:
            assert ud is None
            skip = self.buf(n, hh, ww, cout)
            ask = self.igemm(p + ".skip_connection", t0, c0, skip, cout,
                             self.pack([p + ".skip_connection.weight"], 1), x1=t1, c1=c1, m=n * hh * ww,
                             bias=P(p + ".skip_connection.bias"))
            res, res_mode = skip, L.RS_NONE
        else:
            assert t1 is None
            res, res_mode = t0, rs
        y = self.buf(n, ho, wo, cout)
        ac2 = self.igemm(p + ".out_layers.3", h1, cout, y, cout, self.pack([p + ".out_layers.3.weight"], 3),
                         conv=(n, ho, wo, ho, wo, 1, L.RS_NONE), pro=L.PRO_AFFINE_NC, silu=1, pa=a2, pb=b2,
                         bias=P(p + ".out_layers.3.bias"), res=res, res_mode=res_mode, stats=True)
        pbox, sbox = C.c_float(0.0), C.c_uint32(0)
        self.drops.append((ac2, pbox, sbox))                  # nn.Dropout sits in front of conv2 (openaimodel.py:272)
        self.tape.append(dict(kind="res", drop_p=pbox, drop_seed=sbox, p=p,
                              srcs=[(t0, c0)] + ([(t1, c1)] if t1 is not None else []), cin=cin,
                              cout=cout, hw_in=(hh, ww), hw_out=(ho, wo), rs=rs, a1=a1, b1=b1, sums1=sums1, h1=h1,
                              a2=a2, b2=b2, sums2=sums2, film_off=self.film_off[p], conv1=ac1, conv2=ac2, skip=ask,
                              y=y, ss=ss))
        return (y, cout, ho, wo)

    # ---- execution
    def refresh(self, stream):
        if getattr(self, "_pack_batch", None) is None or len(self._pack_batch.packs) != len(self.packed):
            self._pack_batch = _PackBatch(self.packed, self.prec, self.dev)
        self._pack_batch.refresh(stream)
        for a, pk in self._late:
            a.cin_p, a.cout_p = pk.cin_p, pk.cout_p
        for hook in self.refresh_hooks:
            hook(stream)
        sig = tuple((b.data_ptr(), b._version) for b in self._film_bias_src)
        if sig != getattr(self, "_film_sig", None):
            self._film_bias.copy_(torch.cat([b.detach().reshape(-1) for b in self._film_bias_src]))
            self._film_sig = sig

    # ---- balanced-tail health word (include/sgdm_hip.h: sgd_igemm_work_status_offset).  A finisher whose producers did not
    # arrive within its bounded poll poisons its tile with NaN and raises the word; from then on every split tile on this
    # workspace is poisoned until the host zeroes it.  The paths that own a workspace honour that contract here: the
    # samplers at the end of a trajectory (check_health: one 4-byte read), the training step without a sync of its own
    # (note_health after the backward program, looked at when the next step is prepared).
    def _health_fail(self, training=False):
        self.work.zero_()
        what = ("sgdm_amd: a conv launch's balanced tail timed out waiting for another block's partial sums (stale arrival "
                "counters: a faulted launch, or two streams on one engine); the affected outputs were poisoned with NaN.  The "
                "workspace has been re-zeroed")
        if not training:
            raise RuntimeError(what + ": rerun the step.")
        grad_health(self.dev).zero_()
        raise RuntimeError(what + ".  The gradients of that training iteration are invalid on every rank (the flag travels "
                           "with the gradient exchange).  sgdm_amd.optim.FusedAdamWEma skipped its step for them -- parameters, "
                           "moments and EMA shadows are untouched: repeat the iteration.  Any other optimizer has consumed "
                           "non-finite gradients: restore the last checkpoint.")

    def check_health(self):
        """synchronising check (end of a sampling trajectory, tests)"""
        if self.work_bytes:
            off = int(self.lib.sgd_igemm_work_status_offset()) // 4
            if int(self.work.view(torch.int32)[off].item()) != 0:
                self._health_fail()

    def health_word(self):
        """the workspace's health word as a 1-element int32 view (None without a workspace)"""
        if not self.work_bytes:
            return None
        if getattr(self, "_health_off", None) is None:
            self._health_off = int(self.lib.sgd_igemm_work_status_offset()) // 4
        return self.work.view(torch.int32)[self._health_off:self._health_off + 1]

    def note_health(self, flag=None):
        """asynchronous: copy the word -- or `flag`, the float the training step derived from it and summed over the ranks
        (train.Backward.run), so that every rank of a data-parallel job sees the same value -- to pinned host memory
        behind the launches issued so far"""
        if not self.work_bytes:
            return
        if getattr(self, "_health_host", None) is None:
            self._health_host = torch.zeros(1, dtype=torch.float32).pin_memory()
        self._health_host.copy_(flag if flag is not None else self.health_word(), non_blocking=True)
        self._health_training = flag is not None
        self._health_ev = torch.cuda.Event()
        self._health_ev.record()

    def poll_health(self):
        """look at the last note_health() if its copy has landed (never waits: a training step that is found out one
        iteration later has still not been applied -- the fused optimizer is gated by the same flag on the device)"""
        ev = getattr(self, "_health_ev", None)
        if ev is not None and ev.query():
            self._health_ev = None
            if float(self._health_host[0]) != 0:
                self._health_fail(training=getattr(self, "_health_training", False))

    def set_grid_cap(self, reserve):
        """sgd_igemm_args.grid_cap of every conv / linear launch of the BACKWARD program: all CUs but `reserve` (0: the
        whole device).  The gradient exchange runs under the backward only -- the optimizer waits for it, so the next
        forward finds the device empty again and keeps the whole-device grid (a reserve there would cost its CU share
        of every launch for nothing).  Host-side field of the argument blocks -- a captured hipGraph keeps the value it
        was captured with, which is why only the (never captured) training programs use a reserve."""
        cap = 0 if reserve <= 0 else max(8, device_cus(self.dev) - int(reserve))
        if cap == getattr(self, "_grid_cap", 0):
            return
        self._grid_cap = cap
        bw = getattr(self, "backward", None)
        if bw is not None:
            bw.apply_grid_cap()

    def run(self, x, t, cond, layout, mask, train=False):
        stream = torch.cuda.current_stream().cuda_stream
        self.prepare(x, t, cond, layout, mask, train=train)
        self.launch(stream)

    def prepare(self, x, t, cond, layout, mask, train=False):
        """host side of one evaluation: weight-pack refresh, dropout seeds, input validation / dtype normalisation.
        The tensors are kept (``_keep_inputs``) and read by ``launch`` -- a captured launch sequence (hipGraph) re-reads
        the SAME buffers on every replay, so a caller that replays must update them in place."""
        m, n = self.m, self.n
        stream = torch.cuda.current_stream().cuda_stream
        self.poll_health()
        self.refresh(stream)
        # training step of a data-parallel job: the persistent conv grid leaves CUs to the gradient exchange that runs on a
        # side stream under the backward (ddp.reserved_cus); every other evaluation owns the device
        from .ddp import reserved_cus
        self.set_grid_cap(reserved_cus(m) if train else 0)
        p_drop = float(m.dropout) if (train and m.dropout) else 0.0      # (grad mode is off inside autograd.Function)
        if p_drop != self._p_drop or p_drop > 0:
            seeds = torch.randint(0, 2 ** 31 - 1, (len(self.drops),)).tolist() if p_drop > 0 else [0] * len(self.drops)
            for (a, pbox, sbox), sd in zip(self.drops, seeds):
                a.drop_p, a.drop_seed = p_drop, sd
                pbox.value, sbox.value = p_drop, sd
            self._p_drop = p_drop
        B = x.shape[0]
        assert n % B == 0
        x = x.contiguous().float()
        t = t.contiguous().to(torch.int64)
        mask_u8 = mask.contiguous().view(torch.uint8) if mask is not None else None
        cl = m._in_ch_total - m.in_channels
        lfmt = 0
        if cl:
            if layout is None:
                raise ValueError(f"condition_method={m.condition_method} needs `layout`")
            # compact guidance, expanded on the device (sgdm_amd/guidance.py): a uint8 label map [B,H,W] / [B,1,H,W]
            # instead of the one-hot mask [B,L,H,W], or int box corners [B,4] instead of the box mask [B,1,H,W]
            if layout.dtype == torch.uint8 and layout.numel() == B * self.h * self.w:
                layout, lfmt = layout.contiguous(), 1
            elif layout.dtype in (torch.int32, torch.int64) and tuple(layout.shape) == (B, 4):
                if cl != 1:
                    raise ValueError("box-corner layouts need layout_dim == 1")
                layout, lfmt = layout.to(torch.int32).contiguous(), 2
            else:
                layout = layout.contiguous().float()
                assert layout.shape == (B, cl, self.h, self.w), (layout.shape, (B, cl, self.h, self.w))
        is_i64 = 0
        if m._cond_width:
            if cond is None:
                raise ValueError(f"condition_method={m.condition_method} needs `cond`")
            cond = cond.contiguous()
            if cond.dtype == torch.int64 and cond.numel() == B and m._cond_width > 1:
                is_i64 = 2                                   # class / cluster ids: one-hot expanded on the device
                cond = cond.reshape(B)
            else:
                is_i64 = int(cond.dtype == torch.int64)
                if not is_i64:
                    cond = cond.float()
                assert cond.numel() == B * m._cond_width, (cond.shape, m._cond_width)
        self._keep_inputs = (x, t, cond, layout, mask_u8, B, is_i64, lfmt)
        self._train_mode = bool(train)               # the weight gradient of mlp_cond.0 reads the expanded rows

    def launch(self, stream):
        """device side: boundary kernels + the static launch program (no allocation, no sync: graph-capturable)"""
        m, n, lib = self.m, self.n, self.lib
        x, t, cond, layout, mask_u8, B, is_i64, lfmt = self._keep_inputs
        cl = m._in_ch_total - m.in_channels
        L.check(lib.sgd_timestep_embedding(_ptr(t), _ptr(self.freqs), B, n, m.model_channels, _ptr(self.temb),
                                           stream), "temb")
        if lfmt:
            L.check(lib.sgd_pack_input_compact(_ptr(x), _ptr(layout), lfmt, _ptr(mask_u8), _ptr(m.null_layout_emb), B, n,
                                               m.in_channels, cl, self.h, self.w, _ptr(self.x_in), stream), "pack_input")
        else:
            L.check(lib.sgd_pack_input(_ptr(x), _ptr(layout) if cl else C.c_void_p(0), _ptr(mask_u8),
                                       _ptr(m.null_layout_emb) if cl else C.c_void_p(0), B, n, m.in_channels, cl,
                                       self.h, self.w, _ptr(self.x_in), stream), "pack_input")
        self.cond_ids = self.cond_rows = None
        if m._cond_width:
            if is_i64 == 2 and getattr(self, "gather_cond", False) and not self._train_mode:
                self.cond_ids = (cond, mask_u8, B)           # mlp_cond.0 runs as a column gather (see _build_cond_path)
            elif is_i64 == 1 and getattr(self, "sparse_cond", False):
                self.cond_rows = (cond, mask_u8, B)          # ... or over the non-zero entries of the int64 one-hot rows
                if self._train_mode:                         # (the weight gradient of mlp_cond.0 still reads the expanded rows)
                    L.check(lib.sgd_cond_select(_ptr(cond), int(is_i64), _ptr(mask_u8), _ptr(m.null_cond_emb), B, n,
                                                m._cond_width, _ptr(self.cond_m), stream), "cond_select")
            else:
                L.check(lib.sgd_cond_select(_ptr(cond), int(is_i64), _ptr(mask_u8), _ptr(m.null_cond_emb), B, n,
                                            m._cond_width, _ptr(self.cond_m), stream), "cond_select")
        self.prog.run(stream)
        self.ran = True


# ------------------------------------------------------------------------------------------------
# unet_fast
# ------------------------------------------------------------------------------------------------
class UNetModel(UNetModelBase):
    """``dynamic.diffusionmodules.openaimodel.UNetModel`` (config/dynamic/unet_fast.yaml)."""
    KIND = "unet_fast"

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, legacy=True, cond_dim=None, condition=None, condition_method=None):
        super().__init__()
        if use_spatial_transformer:
            assert context_dim is not None, "Fool!! You forgot to include the dimension of your cross-attention conditioning..."
        if context_dim is not None:                                             # openaimodel.py:527-537
            assert use_spatial_transformer, "Fool!! You forgot to use the spatial transformer for your cross-attention conditioning..."
            context_dim = list(context_dim) if not isinstance(context_dim, int) else context_dim
        self.use_spatial_transformer = use_spatial_transformer
        self.transformer_depth = transformer_depth
        self.context_dim = context_dim
        # QKVAttention instead of QKVAttentionLegacy (openaimodel.py:350-355): q | k | v are split before the heads -- for the
        # attention core a matter of strides (round 6; with the SpatialTransformer the flag has no effect, as in the reference)
        self.use_new_attention_order = bool(use_new_attention_order)
        if condition_method == "cluster_lookup":
            raise NotImplementedError("cluster_lookup (nn.Embedding(888888888, .)) is not supported")
        self._setup(image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                    dropout, channel_mult, conv_resample, dims, use_checkpoint, use_fp16, num_heads,
                    num_head_channels, num_heads_upsample, use_scale_shift_norm, resblock_updown, cond_dim,
                    condition, condition_method)
        ted = 4 * model_channels
        cd = self.cond_dim
        self.mlp_cond_out = ted // 2 if cd > 0 else 0
        self._emb_ch = ted + self.mlp_cond_out
        self._cond_width = cd
        ld = 0
        sp = _Spec()
        if cd > 0:
            sp.items.append(("null_cond_emb", (1, cd), "frozen", _zeros))
        if condition_method == "clusterlayout":
            sp.items.append(("null_layout_emb", (1, 1, image_size, image_size), "frozen", _zeros))
            ld = _layout_dim(condition, "clusterlayout")
        self._in_ch_total = in_channels + ld
        sp.linear("time_embed.0", model_channels, ted)
        sp.linear("time_embed.2", ted, ted)
        if cd > 0:
            sp.linear("mlp_cond.0", cd, ted // 2)
            sp.linear("mlp_cond.2", ted // 2, ted // 2)
        self._register_all(sp)

    def _heads_for(self, ch, up=False):
        if self.use_spatial_transformer and self.num_head_channels == -1:
            return self.num_heads                   # SpatialTransformer(ch, num_heads, ...) in every position (openaimodel.py:683,803)
        return super()._heads_for(ch, up)

    def _attn_spec(self, sp, p, ch, heads):
        if self.use_spatial_transformer:
            return self._st_spec(sp, p, ch, heads)
        sp.norm(p + ".norm", ch)
        sp.conv(p + ".qkv", ch, 3 * ch, 1, dims=1)
        sp.conv(p + ".proj_out", ch, ch, 1, dims=1, zero=True)                  # openaimodel.py:357

    # ---- SpatialTransformer (dynamic/attention.py:238-270; registration order of the reference modules)
    def _st_spec(self, sp, p, ch, heads):
        d = ch // heads
        inner = heads * d
        cd = self.context_dim
        sp.norm(p + ".norm", ch)
        sp.conv(p + ".proj_in", ch, inner, 1)
        for i in range(self.transformer_depth):
            b = f"{p}.transformer_blocks.{i}"
            for att, kdim in ((".attn1", inner), (".ff", None), (".attn2", cd)):
                if att == ".ff":
                    sp.linear(b + ".ff.net.0.proj", inner, inner * 8)           # GEGLU(dim, 4*dim): Linear(dim, 8*dim)
                    sp.linear(b + ".ff.net.2", inner * 4, inner)
                    continue
                sp.linear(b + att + ".to_q", inner, inner, bias=False)
                sp.linear(b + att + ".to_k", kdim, inner, bias=False)
                sp.linear(b + att + ".to_v", kdim, inner, bias=False)
                sp.linear(b + att + ".to_out.0", inner, inner)
            for nm in (".norm1", ".norm2", ".norm3"):
                sp.norm(b + nm, inner)
        sp.conv(p + ".proj_out", inner, ch, 1, zero=True)                        # zero_module (attention.py:254-258)

    def _build_st(self, eng, p, layer, src):
        """SpatialTransformer.forward with context=None (openaimodel.py:915 never builds one: both attentions of a block
        are self-attentions, which requires context_dim == inner_dim exactly as in the reference).  Per block:
        GroupNorm(eps 1e-6) -> 1x1 proj_in -> depth x [LN -> MHA -> +x ; LN -> MHA -> +x ; LN -> GEGLU FF -> +x] -> 1x1
        proj_out -> + x_in, on the existing kernels: GN / LN folded into the GEMM loaders, q|k|v as ONE GEMM, the MFMA
        attention core, bias + residual in the GEMM epilogues, one elementwise launch for the GEGLU gate."""
        _, ch, heads = layer
        t, c, hh, ww = src
        n, T, P, lib = eng.n, hh * ww, self.P, eng.lib
        d = ch // heads
        inner = heads * d
        if self.context_dim != inner:
            raise ValueError(f"use_spatial_transformer with context=None needs context_dim == {inner} (to_k / to_v are applied "
                             f"to the {inner}-wide hidden states, openaimodel.py:915, attention.py:174-176)")
        rows = n * T
        a, b = eng.gn(p + ".norm", [(t, c)], T, p + ".norm", eps=1e-6)
        x = eng.buf(n, T, inner)
        eng.igemm(p + ".proj_in", t, c, x, inner, eng.pack([p + ".proj_in.weight"], 1), m=rows, rows_per_n=T,
                  pro=L.PRO_AFFINE_NC, pa=a, pb=b, bias=P(p + ".proj_in.bias"))
        for i in range(self.transformer_depth):
            blk = f"{p}.transformer_blocks.{i}"
            for att, nrm in ((".attn1", ".norm1"), (".attn2", ".norm2")):
                st = eng.buf(rows, 2)
                eng.prog.add(blk + nrm, lib.sgd_ln_stats, _ptr(x), rows, inner, LN_EPS, _ptr(st))
                qkv = eng.buf(n, T, 3 * inner)
                eng.igemm(blk + att + ".qkv", x, inner, qkv, 3 * inner,
                          eng.pack([blk + att + ".to_q.weight", blk + att + ".to_k.weight", blk + att + ".to_v.weight"], 1),
                          m=rows, pro=L.PRO_LN_ROW, pa=st, pb=P(blk + nrm + ".weight"), pc=P(blk + nrm + ".bias"))
                o = eng.buf(n, T, inner)
                lse = eng.buf(n, heads, T)
                eng.prog.add(blk + att + ".attn", eng.attention_fn(), _ptr(qkv), 3 * inner, d,
                             C.c_void_p(qkv.data_ptr() + 4 * inner), C.c_void_p(qkv.data_ptr() + 8 * inner), 3 * inner, d,
                             n, heads, T, T, d, d ** -0.5, _ptr(o), inner, _ptr(lse))
                xn = eng.buf(n, T, inner)
                eng.igemm(blk + att + ".to_out.0", o, inner, xn, inner, eng.pack([blk + att + ".to_out.0.weight"], 1), m=rows,
                          bias=P(blk + att + ".to_out.0.bias"), res=x)
                x = xn
            st = eng.buf(rows, 2)
            eng.prog.add(blk + ".norm3", lib.sgd_ln_stats, _ptr(x), rows, inner, LN_EPS, _ptr(st))
            hid = eng.buf(rows, 8 * inner)
            eng.igemm(blk + ".ff.net.0.proj", x, inner, hid, 8 * inner, eng.pack([blk + ".ff.net.0.proj.weight"], 1), m=rows,
                      pro=L.PRO_LN_ROW, pa=st, pb=P(blk + ".norm3.weight"), pc=P(blk + ".norm3.bias"),
                      bias=P(blk + ".ff.net.0.proj.bias"))
            gated = eng.buf(rows, 4 * inner)
            eng.prog.add(blk + ".ff.geglu", lib.sgd_geglu, _ptr(hid), rows, 4 * inner, _ptr(gated))
            xn = eng.buf(n, T, inner)
            eng.igemm(blk + ".ff.net.2", gated, 4 * inner, xn, inner, eng.pack([blk + ".ff.net.2.weight"], 1), m=rows,
                      bias=P(blk + ".ff.net.2.bias"), res=x)
            x = xn
        y = eng.buf(n, hh, ww, ch)
        eng.igemm(p + ".proj_out", x, inner, y, ch, eng.pack([p + ".proj_out.weight"], 1), m=rows, rows_per_n=T,
                  bias=P(p + ".proj_out.bias"), res=t, stats=True)
        return (y, ch, hh, ww)

    def _res_prefixes(self):
        return _res_prefixes(self._plan)

    def _build_cond_path(self, eng, emb):
        if self.cond_dim <= 0:
            return
        ted = 4 * self.model_channels
        P = self.P
        c1 = eng.buf(eng.n, ted // 2)
        eng.emb_c = eng.buf(eng.n, ted // 2)                                      # emb = cat(time, cond), :942
        skinny = self.cond_dim >= 1024 and eng.n <= 256          # cluster k = 5000: split the long reduction over blocks
        a0 = eng.igemm("mlp_cond.0", eng.cond_m, self.cond_dim, c1, ted // 2, eng.pack(["mlp_cond.0.weight"], 1),
                       m=eng.n, bias=P("mlp_cond.0.bias"), launch=not skinny)
        if skinny:
            ksplit = max(1, min(64, self.cond_dim // 128))
            work = eng.buf(ksplit, eng.n, ted // 2)
            wt, bs = P("mlp_cond.0.weight"), P("mlp_cond.0.bias")
            lib, n, nout, K = eng.lib, eng.n, ted // 2, self.cond_dim
            nullproj = eng.buf(nout)
            nwork = eng.buf(ksplit, 1, nout)
            eng.gather_cond = True
            eng.sparse_cond = 8 * K <= 64 * 1024 - 2048 and os.environ.get("SGDM_SPARSE_COND", "1") != "0"
            box = dict(sig=None)

            def mlp_cond0(stream):
                """dense skinny GEMM for one-hot / float cond rows; for class / cluster IDS the same result as a column
                gather of the weight (sgd_linear_gather), dropped rows taking the projection of the null embedding"""
                if eng.cond_rows is not None:        # int64 one-hot rows (the reference's dataset format): non-zero entries only
                    rows_, mask_u8, B = eng.cond_rows
                    return lib.sgd_linear_sparse_rows(_ptr(rows_), _ptr(mask_u8), _ptr(wt), _ptr(bs), _ptr(nullproj), B, n, nout,
                                                      K, _ptr(c1), nout, stream)
                if eng.cond_ids is None:
                    return lib.sgd_linear_splitk(_ptr(eng.cond_m), K, _ptr(wt), _ptr(bs), n, nout, K, _ptr(work), ksplit,
                                                 _ptr(c1), nout, stream)
                ids, mask_u8, B = eng.cond_ids
                return lib.sgd_linear_gather(_ptr(ids), _ptr(mask_u8), _ptr(wt), _ptr(bs), _ptr(nullproj), B, n, nout, K,
                                             _ptr(c1), nout, stream)
            mlp_cond0.__name__ = "sgd_linear_splitk"

            def refresh_nullproj(stream):
                """w . null_cond_emb + b (the projection the dropped rows of the id path take), recomputed when a weight
                changes.  It runs from Engine.refresh -- OUTSIDE the launch program -- so a hipGraph-captured step, which
                holds only the gather, reads the current projection after an optimizer step / EMA swap / load_state_dict
                (begin() of a captured trajectory calls refresh)."""
                srcs = (wt, bs, self.null_cond_emb)
                sig = tuple((s_.data_ptr(), s_._version) for s_ in srcs)
                if sig != box["sig"]:
                    L.check(lib.sgd_linear_splitk(_ptr(self.null_cond_emb), K, _ptr(wt), _ptr(bs), 1, nout, K, _ptr(nwork),
                                                  ksplit, _ptr(nullproj), nout, stream), "sgd_linear_splitk(null)")
                    box["sig"] = sig
            eng.refresh_hooks.append(refresh_nullproj)
            eng.prog.add("mlp_cond.0", mlp_cond0, flops=2.0 * n * K * nout, nbytes=4.0 * (n * K + K * nout + n * nout))
            eng.prog.keep.append((wt, bs))
        a2 = eng.igemm("mlp_cond.2", c1, ted // 2, eng.emb_c, ted // 2, eng.pack(["mlp_cond.2.weight"], 1), m=eng.n,
                       silu=1, bias=P("mlp_cond.2.bias"))
        eng.tape.append(dict(kind="mlp2", name="mlp_cond", x=eng.cond_m, h=c1, y=eng.emb_c, a0=a0, a2=a2,
                             cin=self.cond_dim, mid=ted // 2, cout=ted // 2, add_to_y=False))

    def _build_attn(self, eng, p, layer, src):
        """AttentionBlock + QKVAttentionLegacy (openaimodel.py:365-371, 403-420)"""
        if self.use_spatial_transformer:
            return self._build_st(eng, p, layer, src)
        _, ch, heads = layer
        t, c, hh, ww = src
        n, T, P = eng.n, hh * ww, self.P
        d = ch // heads
        # head widths the attention core has no instance for (config/dynamic/unet.yaml: 256 / 32 = 8 channels per head at ds 2) run
        # zero-padded to dp, like Attention_LR's: qkv's packed weight / bias and proj_out's packed weight are padded re-layouts of
        # the parameters (_Pad), qkv and att are dp wide per head, the softmax scale stays that of the TRUE width
        dp = padded_head_dim(d)
        inner = heads * dp
        new_order = getattr(self, "use_new_attention_order", False)
        pads = None
        if dp != d:
            # legacy layout: channel = head*3d + {q: 0, k: d, v: 2d}; new order (QKVAttention): channel = {q: 0, k: ch, v: 2ch} + head*d
            if new_order:
                qkvmap = [s * inner + h * dp + i for s in range(3) for h in range(heads) for i in range(d)]
            else:
                qkvmap = [h * 3 * dp + s * dp + i for h in range(heads) for s in range(3) for i in range(d)]
            omap = [h * dp + i for h in range(heads) for i in range(d)]
            pads = dict(qkv=_Pad(rows=qkvmap, n_rows=3 * inner), qkv_bias=_Pad(cols=qkvmap, n_cols=3 * inner),
                        out=_Pad(cols=omap, n_cols=inner))
        pad = (lambda k: pads[k]) if pads else (lambda k: None)
        a, b = eng.gn(p + ".norm", [(t, c)], T, p + ".norm")
        sums = eng._last_sums
        qkv = eng.buf(n, T, 3 * inner)
        qbias = eng.padded(p + ".qkv.bias", pads["qkv_bias"]) if pads else P(p + ".qkv.bias")
        aq = eng.igemm(p + ".qkv", t, c, qkv, 3 * inner, eng.pack([p + ".qkv.weight"], 1, pad("qkv")), m=n * T, rows_per_n=T,
                       pro=L.PRO_AFFINE_NC, pa=a, pb=b, bias=qbias)
        att = eng.buf(n, T, inner)
        lse = eng.buf(n, heads, T)                         # softmax statistics kept for the backward
        # head stride, k / v offsets of the (padded) layout; scale = (d^-1/4)^2 applied to q.k
        hs, ko, vo = (dp, inner, 2 * inner) if new_order else (3 * dp, dp, 2 * dp)
        eng.prog.add(p + ".attn", eng.attention_fn(), _ptr(qkv), 3 * inner, hs,
                     C.c_void_p(qkv.data_ptr() + 4 * ko), C.c_void_p(qkv.data_ptr() + 4 * vo), 3 * inner, hs,
                     n, heads, T, T, dp, 1.0 / math.sqrt(d), _ptr(att), inner, _ptr(lse))
        y = eng.buf(n, hh, ww, ch)
        ap = eng.igemm(p + ".proj_out", att, inner, y, ch, eng.pack([p + ".proj_out.weight"], 1, pad("out")), m=n * T,
                       rows_per_n=T, bias=P(p + ".proj_out.bias"), res=t, stats=True)
        eng.tape.append(dict(kind="attn", p=p, x=t, ch=ch, heads=heads, d=d, dp=dp, pads=pads, T=T, hw=(hh, ww), a=a, b=b,
                             sums=sums, qkv=qkv, att=att, lse=lse, qkv_args=aq, proj_args=ap, y=y, qkv_layout=(hs, ko, vo)))
        return (y, ch, hh, ww)

    # ---- reference entry points (openaimodel.py:861-956)
    def forward(self, x, timesteps=None, cond=None, layout=None, cond_drop_prob=0.0, image_batch_ids=None,
                cond_drop_mask=None):
        n = len(x)
        if isinstance(cond_drop_prob, (float, int)):
            cond_drop_prob = torch.full((n,), cond_drop_prob, dtype=torch.float, device=x.device)
        assert isinstance(cond_drop_prob, torch.Tensor)
        mask = None
        if self.cond_dim > 0:
            mask = cond_drop_mask if cond_drop_mask is not None else self._draw_mask(n, cond_drop_prob, x.device)
        eng = self._run(x, timesteps, cond, layout, mask, n)
        if isinstance(eng, tuple):          # training path returns tensors directly
            return eng
        return self._to_nchw(eng), 0.0, dict()

    def forward_with_cond_scale(self, x, t, cond_scale, cond, layout=None, p0=None, image_batch_ids=None):
        B = x.shape[0]
        if p0 is None:
            p0 = torch.full((B,), 0.0, dtype=torch.float, device=x.device)
        p1 = torch.full((B,), 1.0, dtype=torch.float, device=x.device)
        is_num = isinstance(cond_scale, (int, float))
        if is_num and cond_scale == 1:
            return self.forward(x=x, timesteps=t, cond_drop_prob=p0, cond=cond, layout=layout)[0]
        if is_num and cond_scale == 0:
            return self.forward(x=x, timesteps=t, cond_drop_prob=p1, cond=cond, layout=layout)[0]
        return _cfg_eval(self, x, t, cond_scale, cond, layout, torch.cat((p0, p1), 0))


# ------------------------------------------------------------------------------------------------
# unetca_fast
# ------------------------------------------------------------------------------------------------
class UNetModelCA(UNetModelBase):
    """``dynamic.diffusionmodules.openaimodel_ca.UNetModel`` (config/dynamic/unetca_fast.yaml)."""
    KIND = "unetca_fast"

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_ca_block=False, transformer_depth=1, context_dim=None,
                 n_embed=None, legacy=True, cond_token_num=0, cond_dim=None, use_cls_token_as_pooled=None,
                 condition=None, condition_method=None):
        super().__init__()
        if not use_ca_block:
            raise NotImplementedError("openaimodel_ca without use_ca_block is not a shipped configuration")
        assert isinstance(cond_dim, int) and cond_token_num >= 0                 # openaimodel_ca.py:559-560
        if cond_token_num == 0:
            assert cond_dim == 0                                                 # :563
        self._setup(image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                    dropout, channel_mult, conv_resample, dims, use_checkpoint, use_fp16, num_heads,
                    num_head_channels, num_heads_upsample, use_scale_shift_norm, resblock_updown, cond_dim,
                    condition, condition_method)
        self.num_classes = num_classes
        self.cond_token_num = cond_token_num
        self.context_dim = context_dim
        self.use_cls_token_as_pooled = use_cls_token_as_pooled
        mc = model_channels
        ted = 4 * mc
        cd = self.cond_dim
        self._emb_ch = ted
        # cond rows as the boundary kernels see them: [cd] (one guidance vector) or the flattened [T, cd] token matrix
        self._cond_width = cd * cond_token_num
        sp = _Spec()
        if cond_token_num >= 1:
            sp.items.append(("null_cond_emb", (cond_token_num, cd), "frozen", _zeros))      # openaimodel_ca.py:566-575
        ld = 0
        if condition_method in ("clusterlayout", "stegoclusterlayout", "layout"):
            sp.items.append(("null_layout_emb", (1, 1, image_size, image_size), "frozen", _zeros))
            ld = _layout_dim(condition, condition_method)
        self._in_ch_total = in_channels + ld
        sp.linear("time_embed.0", mc, ted)
        sp.linear("time_embed.2", ted, ted)
        sp.norm("norm_cond", context_dim)
        sp.linear("to_time_tokens.0", mc, mc)
        sp.linear("to_time_tokens.2", mc, context_dim * NUM_TIME_TOKENS)
        if cond_token_num > 0:
            sp.linear("cond_mlp.0", cd, ted)
            sp.linear("cond_mlp.2", ted, ted)
            sp.linear("to_cond_tokens.0", cd, context_dim * NUM_COND_TOKENS)
            mid = int(math.sqrt(context_dim * cd))
            sp.linear("to_cond_tokens_2d.0", cd, mid)
            sp.linear("to_cond_tokens_2d.2", mid, mid)
            sp.linear("to_cond_tokens_2d.4", mid, mid)
            sp.linear("to_cond_tokens_2d.6", mid, context_dim)
        self._register_all(sp)

    def _attn_spec(self, sp, p, ch, heads):
        dh = ch // heads
        cd = self.context_dim
        sp.items.append((p + ".null_kv", (2, dh), "param", _randn))
        sp.items.append((p + ".norm.gamma", (ch,), "param", _ones))
        sp.items.append((p + ".norm.beta", (ch,), "buffer", _zeros))
        sp.linear(p + ".to_q", ch, dh * heads, bias=False)
        sp.linear(p + ".to_kv", ch, 2 * dh, bias=False)
        sp.norm(p + ".to_context.0", cd)
        sp.linear(p + ".to_context.1", cd, 2 * dh)
        sp.linear(p + ".to_out.0", dh * heads, ch, bias=False)
        sp.items.append((p + ".to_out.1.gamma", (ch,), "param", _ones))
        sp.items.append((p + ".to_out.1.beta", (ch,), "buffer", _zeros))

    def _res_prefixes(self):
        return _res_prefixes(self._plan)

    def _build_cond_path(self, eng, emb):
        """openaimodel_ca.py:942-1017: time tokens, cond tokens, emb += cond_mlp(cond), context LayerNorm"""
        n, P = eng.n, self.P
        mc, ctx = self.model_channels, self.context_dim
        ted = 4 * mc
        T = self.cond_token_num
        ntok = NUM_TIME_TOKENS + (NUM_COND_TOKENS if T == 1 else (T if T > 1 else 0))
        raw = eng.buf(n, ntok, ctx)
        raw2 = raw.view(n, ntok * ctx)
        wt = ctx * NUM_TIME_TOKENS
        raw_t, raw_c = raw2[:, :wt], raw2[:, wt:]           # strided views: keys of the token gradients
        t1 = eng.buf(n, mc)
        a0 = eng.igemm("to_time_tokens.0", eng.temb, mc, t1, mc, eng.pack(["to_time_tokens.0.weight"], 1), m=n,
                       bias=P("to_time_tokens.0.bias"))
        a2 = eng.igemm("to_time_tokens.2", t1, mc, raw, wt, eng.pack(["to_time_tokens.2.weight"], 1),
                       m=n, silu=1, bias=P("to_time_tokens.2.bias"), y_ld=ntok * ctx)
        eng.tape.append(dict(kind="mlp2", name="to_time_tokens", x=eng.temb, h=t1, y=raw_t, a0=a0, a2=a2, cin=mc,
                             mid=mc, cout=wt, add_to_y=False))
        if self.cond_token_num == 1:
            al = eng.igemm("to_cond_tokens.0", eng.cond_m, self.cond_dim, raw, ctx * NUM_COND_TOKENS,
                           eng.pack(["to_cond_tokens.0.weight"], 1), m=n, bias=P("to_cond_tokens.0.bias"),
                           y_ld=ntok * ctx, y_off=wt)
            eng.tape.append(dict(kind="linear", name="to_cond_tokens.0", y=raw_c, a=al, cin=self.cond_dim,
                                 cout=ctx * NUM_COND_TOKENS))
            c1 = eng.buf(n, ted)
            b0 = eng.igemm("cond_mlp.0", eng.cond_m, self.cond_dim, c1, ted, eng.pack(["cond_mlp.0.weight"], 1), m=n,
                           bias=P("cond_mlp.0.bias"))
            b2 = eng.igemm("cond_mlp.2", c1, ted, emb, ted, eng.pack(["cond_mlp.2.weight"], 1), m=n, silu=1,
                           bias=P("cond_mlp.2.bias"), res=emb)                    # emb = emb + cond_condensed, :977
            eng.tape.append(dict(kind="mlp2", name="cond_mlp", x=eng.cond_m, h=c1, y=emb, a0=b0, a2=b2,
                                 cin=self.cond_dim, mid=ted, cout=ted, add_to_y=True))
        if T > 1:
            # token guidance (openaimodel_ca.py:987-1013): cond [n, T, cd] -> per-token MLP to_cond_tokens_2d -> T context
            # tokens behind the 8 time tokens; emb += cond_mlp(pooled token)
            cd = self.cond_dim
            mid = int(math.sqrt(ctx * cd))
            rows = n * T
            src, width = eng.cond_m, cd
            layers = []
            for li, (idx, wout) in enumerate(((0, mid), (2, mid), (4, mid), (6, ctx))):
                name = f"to_cond_tokens_2d.{idx}"
                last = idx == 6
                dst = raw if last else eng.buf(rows, wout)
                al = eng.igemm(name, src, width, dst, wout, eng.pack([name + ".weight"], 1), m=rows, silu=int(li > 0),
                               bias=P(name + ".bias"), orows=(T, ntok, NUM_TIME_TOKENS) if last else (0, 0, 0))
                layers.append(dict(name=name, x=src, cin=width, cout=wout, a=al))
                src, width = dst, wout
            eng.tape.append(dict(kind="mlp_chain", layers=layers, y=raw_c, rows=rows))
            pooled = eng.buf(n, cd)
            eng.prog.add("cond_pooled", eng.lib.sgd_token_pool, _ptr(eng.cond_m), n, T, cd,
                         1 if self.use_cls_token_as_pooled == True else 0, _ptr(pooled))     # noqa: E712  (reference: == True)
            c1 = eng.buf(n, ted)
            b0 = eng.igemm("cond_mlp.0", pooled, cd, c1, ted, eng.pack(["cond_mlp.0.weight"], 1), m=n,
                           bias=P("cond_mlp.0.bias"))
            b2 = eng.igemm("cond_mlp.2", c1, ted, emb, ted, eng.pack(["cond_mlp.2.weight"], 1), m=n, silu=1,
                           bias=P("cond_mlp.2.bias"), res=emb)
            eng.tape.append(dict(kind="mlp2", name="cond_mlp", x=pooled, h=c1, y=emb, a0=b0, a2=b2, cin=cd, mid=ted,
                                 cout=ted, add_to_y=True))
            eng.token_guidance = True
        context = eng.buf(n, ntok, ctx)
        eng.prog.add("norm_cond", eng.lib.sgd_ln_apply, _ptr(raw), _ptr(P("norm_cond.weight")),
                     _ptr(P("norm_cond.bias")), C.c_void_p(0), n * ntok, ctx, LN_EPS, _ptr(context))
        eng.tape.append(dict(kind="norm_cond", raw=raw, raw_t=raw_t, raw_c=raw_c if self.cond_token_num >= 1 else None,
                             context=context, ntok=ntok, ctx=ctx, wt=wt))
        eng.context, eng.ntok = context, ntok

    def _build_attn(self, eng, p, layer, src):
        """Attention_LR.forward (crossattetion_lr.py:81-142).  Head widths the attention core has no instance for run
        zero-padded to dp (padded_head_dim): the projections' packed weights / bias / null_kv are padded re-layouts of the
        parameters (_Pad), every buffer between to_q / to_kv / to_context and to_out is dp wide per head, the softmax
        scale stays dim_head ** -0.5."""
        _, ch, heads = layer
        t, c, hh, ww = src
        n, T, P, lib = eng.n, hh * ww, self.P, eng.lib
        d = ch // heads
        dp = padded_head_dim(d)
        pads = None
        if dp != d:
            qmap = [h * dp + i for h in range(heads) for i in range(d)]          # (head, i) of the inner dimension
            kvmap = list(range(d)) + [dp + i for i in range(d)]                  # [k | v] of the shared key/value head
            pads = dict(q=_Pad(rows=qmap, n_rows=heads * dp), kv=_Pad(rows=kvmap, n_rows=2 * dp),
                        out=_Pad(cols=qmap, n_cols=heads * dp), vec=_Pad(cols=kvmap, n_cols=2 * dp),
                        null=_Pad(cols=list(range(d)), n_cols=dp))
        pad = (lambda k: pads[k]) if pads else (lambda k: None)
        ntok = eng.ntok
        J = ntok + 1 + T                                   # [context | null | self]
        st = eng.buf(n * T, 2)
        eng.prog.add(p + ".norm", lib.sgd_ln_stats, _ptr(t), n * T, c, LN_EPS, _ptr(st))
        q = eng.buf(n, T, heads * dp)
        gamma, beta = P(p + ".norm.gamma"), P(p + ".norm.beta")
        aq = eng.igemm(p + ".to_q", t, c, q, heads * dp, eng.pack([p + ".to_q.weight"], 1, pad("q")), m=n * T,
                       pro=L.PRO_LN_ROW, pa=st, pb=gamma, pc=beta)
        kv = eng.buf(n, J, 2 * dp)
        akv = eng.igemm(p + ".to_kv", t, c, kv, 2 * dp, eng.pack([p + ".to_kv.weight"], 1, pad("kv")), m=n * T,
                        pro=L.PRO_LN_ROW, pa=st, pb=gamma, pc=beta, orows=(T, J, ntok + 1))
        cst = eng.buf(n * ntok, 2)
        eng.prog.add(p + ".to_context.0", lib.sgd_ln_stats, _ptr(eng.context), n * ntok, self.context_dim, LN_EPS,
                     _ptr(cst))
        cbias = eng.padded(p + ".to_context.1.bias", pads["vec"]) if pads else P(p + ".to_context.1.bias")
        actx = eng.igemm(p + ".to_context.1", eng.context, self.context_dim, kv, 2 * dp,
                         eng.pack([p + ".to_context.1.weight"], 1, pad("kv")), m=n * ntok, pro=L.PRO_LN_ROW, pa=cst,
                         pb=P(p + ".to_context.0.weight"), pc=P(p + ".to_context.0.bias"),
                         bias=cbias, orows=(ntok, J, 0))
        null_kv = eng.padded(p + ".null_kv", pads["null"]) if pads else P(p + ".null_kv")
        eng.prog.add(p + ".null_kv", lib.sgd_fill_null_kv, _ptr(null_kv), n, J, ntok, dp, _ptr(kv))
        att = eng.buf(n, T, heads * dp)
        lse = eng.buf(n, heads, T)
        eng.prog.add(p + ".attn", eng.attention_fn(), _ptr(q), heads * dp, dp, _ptr(kv),
                     C.c_void_p(kv.data_ptr() + 4 * dp), 2 * dp, 0, n, heads, T, J, dp, d ** -0.5, _ptr(att),
                     heads * dp, _ptr(lse))
        o = eng.buf(n, T, ch)
        aout = eng.igemm(p + ".to_out.0", att, heads * dp, o, ch, eng.pack([p + ".to_out.0.weight"], 1, pad("out")),
                         m=n * T)
        y = eng.buf(n, hh, ww, ch)
        eng.prog.add(p + ".to_out.1", lib.sgd_ln_apply, _ptr(o), _ptr(P(p + ".to_out.1.gamma")),
                     _ptr(P(p + ".to_out.1.beta")), _ptr(t), n * T, ch, LN_EPS, _ptr(y))
        eng.tape.append(dict(kind="attn_lr", p=p, x=t, ch=ch, heads=heads, d=d, dp=dp, pads=pads, T=T, J=J, ntok=ntok,
                             hw=(hh, ww), q=q, kv=kv, att=att, lse=lse, o=o, aq=aq, akv=akv, actx=actx, aout=aout, y=y,
                             context=eng.context, ctx=self.context_dim))
        return (y, ch, hh, ww)

    # ---- reference entry points (openaimodel_ca.py:879-1033)
    def forward(self, x, timesteps=None, cond_drop_prob=0.0, cond=None, layout=None, cond_drop_mask=None):
        n = len(x)
        if isinstance(cond_drop_prob, (float, int)):
            cond_drop_prob = torch.full((n,), cond_drop_prob, dtype=torch.float, device=x.device)
        else:
            assert isinstance(cond_drop_prob, torch.Tensor)
        mask = None
        if self.cond_token_num == 0:
            if self.condition_method == "clusterlayout":
                raise NotImplementedError                                         # openaimodel_ca.py:947-948
            if self.condition_method == "layout":
                mask = cond_drop_mask if cond_drop_mask is not None else self._draw_mask(n, cond_drop_prob, x.device)
        else:
            if self.cond_token_num > 1:
                assert cond is not None and len(cond.shape) == 3                  # :988  [B, T, C]
                if self.condition_method == "clusterlayout":
                    raise NotImplementedError                                     # :1006-1007
            else:
                assert cond is not None and (len(cond.shape) == 2 or cond.dtype == torch.int64)   # :961 (+ compact ids [B])
            mask = cond_drop_mask if cond_drop_mask is not None else self._draw_mask(n, cond_drop_prob, x.device)
        eng = self._run(x, timesteps, cond, layout, mask, n)
        if isinstance(eng, tuple):
            return eng
        return self._to_nchw(eng), 0.0, dict()

    def forward_with_cond_scale(self, x, t, cond_scale, cond=None, layout=None):
        B = x.shape[0]
        p0 = torch.full((B,), 0.0, dtype=torch.float, device=x.device)
        p1 = torch.full((B,), 1.0, dtype=torch.float, device=x.device)
        if isinstance(cond_scale, int) and cond_scale == 1:                       # int only, as the reference (:882)
            return self.forward(x=x, timesteps=t, cond_drop_prob=p0, cond=cond, layout=layout)[0]
        if isinstance(cond_scale, int) and cond_scale == 0:
            return self.forward(x=x, timesteps=t, cond_drop_prob=p1, cond=cond, layout=layout)[0]
        return _cfg_eval(self, x, t, cond_scale, cond, layout, torch.cat((p0, p1), 0))


# ------------------------------------------------------------------------------------------------
def _cfg_eval(model, x, t, cond_scale, cond, layout, probs):
    """batch-doubled CFG evaluation (openaimodel.py:885-902): rows [0,B) conditional, [B,2B) dropped;
    the doubling is done inside the boundary kernels (row % B), not by torch.cat."""
    B = x.shape[0]
    has_mask = (model._cond_width > 0) or (model._in_ch_total > model.in_channels)
    mask = model._draw_mask(2 * B, probs, x.device) if has_mask else None
    eng = model._run(x, t, cond, layout, mask, 2 * B)
    if torch.is_tensor(cond_scale):
        # per-sample guidance weights [B,1,1,1] (cond-scale sweeps, ddim_plms_sampler.py:117-142): plain broadcasting on
        # the two NCHW halves, exactly get_guided_score (openaimodel.py:853-859)
        eps_c, eps_u = torch.chunk(model._to_nchw(eng), 2, dim=0)
        return model.get_guided_score(eps_u, eps_c, cond_scale.to(eps_c.device))
    if not isinstance(cond_scale, (int, float)):
        raise TypeError(f"cond_scale must be a number or a tensor, got {type(cond_scale)}")
    return model._cfg_combine(eng, cond_scale, B)


def _layout_dim(condition, method):
    node = condition[method] if isinstance(condition, dict) else getattr(condition, method)
    return node["layout_dim"] if isinstance(node, dict) else node.layout_dim


def _res_prefixes(plan):
    inp, mid, out = plan
    names = []
    for i, blk in enumerate(inp):
        names += [(f"input_blocks.{i}.{j}", l[2]) for j, l in enumerate(blk) if l[0] == "res"]
    names += [(f"middle_block.{j}", l[2]) for j, l in enumerate(mid) if l[0] == "res"]
    for i, blk in enumerate(out):
        names += [(f"output_blocks.{i}.{j}", l[2]) for j, l in enumerate(blk) if l[0] == "res"]
    return names
