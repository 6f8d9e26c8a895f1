"""Training step of the drop-in denoisers: backward program + autograd glue.

The reference gets its backward from torch.autograd through the UNet (PL calls ``loss.backward()``,
lightning_module.py:215-245; AttentionBlock recomputes its forward under a custom checkpoint,
openaimodel.py:359-362, util.py:119-148).  Here the backward is a second static launch program
walking the forward tape in reverse:

  * input gradients of every conv / linear = the FORWARD implicit-GEMM kernel on adjoint-packed weights
    (``sgd_pack_weight_dgrad``), in the model's arithmetic mode;
  * weight gradients = ``sgd_wgrad`` (exact-fp32 MFMA, the activated input recomputed by the same fused
    prologue as the forward: nothing but the raw feature maps is kept) + ``sgd_wgrad_reduce`` into the
    reference OIHW layout;
  * GroupNorm(+FiLM)+SiLU backward = ``sgd_gn_bwd_reduce/coef/apply`` (resample adjoints of the up/down
    ResBlocks and the identity-skip gradient folded into the apply pass);
  * attention backward = ``sgd_attention_bwd`` with the forward's log-sum-exp (no score matrix stored);
  * q_sample / MSE loss = ``sgd_q_sample`` / ``sgd_mse_loss``.

``loss.backward()`` works because the UNet evaluation is a ``torch.autograd.Function`` whose inputs are the
trainable parameters; it returns one gradient tensor per parameter in the reference layout.
Both operators are covered: ``unet_fast`` and ``unetca_fast`` (LayerNorm, multi-query attention with context / null
keys, strided-conv Downsample and nearest+conv Upsample adjoints).

One engine (workspace + tape) exists per (UNet batch, H, W, precision), so the activations a backward reads are those of
the LAST forward on that engine: ``_UNetTrainFn.backward`` checks a forward-generation counter and raises if another
forward ran in between (two losses on one model need two backward-before-next-forward passes, as here, or two batches
of different size).
"""
import ctypes as C
import math
import os

import torch

from . import _lib as L
from .unet import GN_EPS, GN_GROUPS, _Program, _ptr


class _PackedAdj:
    """adjoint-packed weight (dgrad operator) of one forward weight, refreshed on version change.
    ``src_fn`` returns the forward weight [cout_fwd, cin_fwd(, k, k)] (may be a slice/cat of parameters)."""

    def __init__(self, deps, src_fn, cout_fwd, cin_fwd, ksize, prec, device):
        self.deps, self.src_fn, self.ksize, self.prec = deps, src_fn, ksize, prec
        self.cout_fwd, self.cin_fwd = cout_fwd, cin_fwd
        lib = L.load()
        nbytes = lib.sgd_packed_weight_bytes(cin_fwd, cout_fwd, ksize, prec)       # adjoint: outputs = cin_fwd
        self.buf = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        self.cin_p = self.cout_p = 0
        self.sig = None
        self.scaled = prec != L.PREC_F32           # per-tensor power-of-two scale, as unet._Packed
        if self.scaled:
            self.amax = torch.zeros(1, dtype=torch.int32, device=device)
            self.scale_inv = torch.ones(1, dtype=torch.float32, device=device)

    @property
    def scale_ptr(self):
        return self.scale_inv.data_ptr() if self.scaled else 0

    def refresh(self, stream):
        sig = tuple((p.data_ptr(), p._version) for p in self.deps)
        if sig == self.sig:
            return
        src = self.src_fn().detach().contiguous().float()
        cin_p, cout_p = C.c_int32(0), C.c_int32(0)
        lib = L.load()
        if self.scaled:
            amax = None
            if len(self.deps) == 1 and src.data_ptr() == self.deps[0].data_ptr():
                # the whole parameter, already reduced by the forward pack at this version (same stream, earlier)
                from .unet import AMAX_OF
                hit = AMAX_OF.get(self.deps[0].data_ptr())
                if hit is not None and hit[0] == self.deps[0]._version and hit[2]() is self.deps[0]:
                    amax = hit[1]
            if amax is None:
                amax = self.amax
                amax.zero_()
                L.check(lib.sgd_weight_amax(_ptr(src), src.numel(), _ptr(amax), stream), "sgd_weight_amax")
            L.check(lib.sgd_pack_weight_scaled(_ptr(src), _ptr(self.buf), self.cout_fwd, self.cin_fwd, self.ksize, self.prec,
                                               1, _ptr(amax), _ptr(self.scale_inv), C.byref(cin_p), C.byref(cout_p),
                                               stream), "sgd_pack_weight_scaled")
        else:
            L.check(lib.sgd_pack_weight_dgrad(_ptr(src), _ptr(self.buf), self.cout_fwd, self.cin_fwd, self.ksize,
                                              self.prec, C.byref(cin_p), C.byref(cout_p), stream), "pack_dgrad")
        self.cin_p, self.cout_p = cin_p.value, cout_p.value
        self._keep = src
        self.sig = sig


def wgrad_ksplit(taps, cout, cin, rows):
    """K split of a weight-gradient launch (sgd_wgrad's grid = co tiles x ci tiles x ksplit).  3x3 convs run as ONE
    512-thread block per CU (csrc/backward.hip: wgrad_conv_ws_kernel, 32 input channels per block): the split is chosen
    so that the grid is as close as possible to a whole number of rounds of the 256 CUs with at least ~8 K tiles per
    block; 1x1 / linear launches (256-thread blocks, two per CU, 128 input channels per block) keep the round-2 rule."""
    ktiles = (rows + 63) // 64
    # stem / head (csrc/backward.hip, wgrad_impl: wgrad_narrow_kernel): 3 / 4 channels on one side and, on the other, a
    # whole number of waves of lanes that divides (or equals) the 256-thread block -- the SAME predicate as the launcher's
    # (`lanes_co` / `lanes_ci`); other widths (the ch = 224 plans) fall through to the generic kernels and their rules
    lanes = lambda c: c % 64 == 0 and (c == 256 or (c < 256 and 256 % c == 0))
    if taps == 9 and ((3 <= cin <= 4 and lanes(cout)) or (cout == 3 and lanes(cin))):
        # HBM-bound row walks, one 256-thread block per slab in chunks of 128 rows -- two blocks per CU at bs 80 (~640
        # rows each)
        return max(1, min(ktiles, 1024, -(-rows // 640)))
    if taps == 9:
        per_k = ((cout + 127) // 128) * ((cin + 31) // 32)
        best, best_cost = 1, None
        for ks in range(1, min(ktiles, 512) + 1):
            grid = per_k * ks
            rounds = -(-grid // 256)
            tiles = -(-ktiles // ks)
            cost = rounds * (tiles + 3.0)              # + ~3 tile periods of prologue / slab store per block
            if tiles >= 4 and (best_cost is None or cost < best_cost - 1e-9):
                best, best_cost = ks, cost
        return best
    base = taps * ((cout + 127) // 128) * ((cin + 127) // 128)
    return max(1, min(ktiles, 1024 // base))


class Backward:
    """backward launch program of one engine (unet_fast)"""

    def __init__(self, eng, arena=None, reducer=None):
        """arena / reducer (sgdm_amd.ddp): parameter gradients are views of one flat buffer cut into buckets and
        each bucket's all-reduce is enqueued on a side stream as soon as the launches filling it are issued."""
        self.e, self.lib, self.n, self.dev, self.prec = eng, eng.lib, eng.n, eng.dev, eng.prec
        self.m = eng.m
        self.prog = _Program()
        self.packs, self.late, self.late_at = [], [], []
        self.arena, self.reducer = arena, reducer
        self.world = reducer.world if reducer is not None else 1
        # Gradients run through the program multiplied by a power of two so that the split-f16 operands of the
        # dgrad launches stay in fp16's normal range (d loss / d eps ~ 1/(B*C*H*W) would land in the subnormals
        # and lose the lo part); every parameter-gradient reduction multiplies by 1/scale again.  Exact in fp32.
        numel = eng.n * self.m.out_channels * eng.h * eng.w
        self.gscale = float(2 ** int(math.ceil(math.log2(max(2, numel)))))
        self.unscale = 1.0 / (self.gscale * self.world)      # + the 1/world of the gradient average (SUM all-reduce)
        self.G = {}                      # activation data_ptr -> [grad tensor, written?]
        self.pgrad = {}                  # parameter name -> gradient tensor (reference shape)
        self.keep = []
        self.CW = 256                    # row chunks of the two-stage column sums
        self.cwork = torch.empty(self.CW, 8192, dtype=torch.float32, device=self.dev)
        self._build()

    # ---------------------------------------------------------------- helpers
    def attention_bwd_fn(self, d):
        """the attention backward in the engine's arithmetic: split-precision MFMA for f16x3 at head widths up to 64 (the
        forward's choice, unet._Engine.attention_fn), exact fp32 otherwise -- 128-wide heads (config/dynamic/unet_fast_s64.yaml:
        1024 channels / 8 heads) included"""
        if self.prec == L.PREC_F16X3 and d <= 64 and os.environ.get("SGDM_ATTN_BWD_SPLIT", "1") != "0":
            return self.lib.sgd_attention_bwd_split
        return self.lib.sgd_attention_bwd

    def buf(self, *shape):
        t = torch.empty(*shape, dtype=torch.float32, device=self.dev)
        self.keep.append(t)
        return t

    def gact(self, t):
        """(grad buffer of activation tensor t, accumulate flag for the next writer)"""
        ent = self.G.get(t.data_ptr())
        if ent is None:
            ent = [self.buf(*t.shape), False]
            self.G[t.data_ptr()] = ent
        acc = ent[1]
        ent[1] = True
        return ent[0], int(acc)

    def gread(self, t):
        ent = self.G[t.data_ptr()]
        assert ent[1], "gradient read before any producer wrote it"
        return ent[0]

    def pg(self, name):
        if name not in self.pgrad:
            self.pgrad[name] = self.arena.grad(name) if self.arena is not None else self.buf(*self.m.P(name).shape)
        return self.pgrad[name]

    def wrote(self, name):
        """program point right after the launch that completes parameter `name`: overlapped bucket send"""
        if self.reducer is None or not self.reducer.active:
            return
        bi = self.arena.bucket_of[name]
        # (the last bucket is flushed by reducer.finish(), after the step's health flag has been written into its tail slot)
        if self.arena.buckets[bi][2] == name and bi != len(self.arena.buckets) - 1:
            red = self.reducer

            def bucket_ready(stream, bi=bi):
                red.bucket_ready(bi)
                return 0
            self.prog.add(f"bucket{bi}.allreduce", bucket_ready)

    def dgrad(self, tag, gy, cin_of_gy, y, cout_of_y, deps, src_fn, cout_fwd, cin_fwd, ksize, conv=None, m=0,
              y_ld=None, zero_up=False, acc=False):
        """y (+)= adjoint(W) applied to gy (gy has cout_fwd channels, y gets cin_fwd channels).
        zero_up: adjoint of a stride-2 conv (gy is zero-upsampled x2 inside the loader); acc: add into y."""
        # the head (a conv with 3 / 4 OUTPUT channels): its input gradient is the stem's kernel on the flipped, transposed
        # parameter (csrc/narrow.hip, adjoint mode) -- fp32 FMA straight from the weight, no adjoint pack
        if (conv is not None and ksize == 3 and not zero_up and not acc and cin_of_gy in (3, 4) and cout_fwd == cin_of_gy
                and cout_of_y % 4 == 0 and cout_of_y <= 1024 and len(deps) == 1 and deps[0].dtype == torch.float32
                and deps[0].is_contiguous() and gy.shape[-1] == cin_of_gy and os.environ.get("SGDM_NARROW_CONV", "1") != "0"):
            nimg, hh, ww = conv
            wsrc, fn, gy_p, y_p, ld = deps[0], self.lib.sgd_conv3_narrow_in, C.c_void_p(gy.data_ptr()), C.c_void_p(y.data_ptr()), y_ld or cout_of_y

            def sgd_conv3_narrow_in(stream):          # (the parameter's address is read at launch time)
                return fn(gy_p, C.c_void_p(wsrc.data_ptr()), C.c_void_p(0), y_p, C.c_void_p(0), nimg, hh, ww, cin_of_gy, cout_of_y,
                          ld, 1, stream)
            self.prog.add(tag, sgd_conv3_narrow_in, flops=2.0 * nimg * hh * ww * cout_of_y * 9 * cin_of_gy)
            return
        pk = _PackedAdj(deps, src_fn, cout_fwd, cin_fwd, ksize, self.prec, self.dev)
        self.packs.append(pk)
        a = L.IgemmArgs()
        a.x0, a.c0 = gy.data_ptr(), cin_of_gy
        if conv is not None:
            nimg, hh, ww = conv                    # dims of gy
            if zero_up:
                a.mode, a.n, a.hi, a.wi, a.ho, a.wo, a.stride, a.resample = (L.MODE_CONV3, nimg, hh, ww, 2 * hh, 2 * ww, 1,
                                                                          L.RS_ZEROUP2)
            else:
                a.mode, a.n, a.hi, a.wi, a.ho, a.wo, a.stride = L.MODE_CONV3, nimg, hh, ww, hh, ww, 1
        else:
            a.mode, a.m, a.stride = L.MODE_FLAT, m, 1
        a.w = pk.buf.data_ptr()
        a.w_scale_inv = pk.scale_ptr
        a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cout_of_y, (y_ld or cout_of_y), self.prec
        if acc:
            a.res, a.res_mode = y.data_ptr(), L.RS_NONE
        if self.e.work_bytes:                # the forward engine's balanced-tail scratch (same stream: launches are ordered)
            a.work, a.work_bytes = self.e.work.data_ptr(), self.e.work_bytes
        a.grid_cap = getattr(self.e, "_grid_cap", 0)
        self.late_at.append(len(self.prog.ops))          # program position of this launch (reserve windows)
        self.late.append((a, pk))
        self.keep.append(a)
        rows = (a.n * a.ho * a.wo) if conv is not None else m
        self.prog.add(tag, self.lib.sgd_igemm, C.byref(a), flops=2.0 * rows * cout_of_y * (9 if conv is not None else 1) * cin_of_gy)

    def wgrad(self, tag, fwd_args, gy, gy_ld, cout, cin, taps, rows, wname, bias_name=None, dw_view=None):
        ktiles = (rows + 63) // 64
        ksplit = wgrad_ksplit(taps, cout, cin, rows)
        slabs = self.buf(ksplit, taps, cout, cin)
        xt = os.environ.get("SGDM_WGRAD_TUNE")      # A/B switch of tools/profile_train_layers.py: "<SGD_TUNE_WGRAD_* bits>[:max cin*cout]"
        if xt and taps == 9 and cin * cout <= int((xt.split(":") + ["1000000000"])[1]):
            fwd_args.tune |= int(xt.split(":")[0])
        # 3x3 convs: operands pre-split once into 16-bit planes (scratch shared by every launch of the program, grown to
        # the largest request before the first run: sgd_wgrad_scratch)
        need = int(self.lib.sgd_wgrad_scratch_bytes(C.byref(fwd_args), cout)) if taps == 9 else 0
        self.wscratch_bytes = max(getattr(self, "wscratch_bytes", 0), need)
        # the bias gradient (column sums of gy) comes out of the same launch: the kernel (or its pre-pass) stages the gy
        # rows anyway; as many partial rows as that launch writes
        brows = int(self.lib.sgd_wgrad_bias_rows(C.byref(fwd_args), cout, gy_ld, ksplit, need)) if bias_name is not None else 0
        bslab = self.buf(brows, cout) if bias_name is not None else None
        box = self

        def wgrad_launch(stream, fwd_args=fwd_args, gy=gy, slabs=slabs, bslab=bslab):
            ws = box._wscratch()
            return box.lib.sgd_wgrad_scratch(C.byref(fwd_args), _ptr(gy), gy_ld, cout, _ptr(slabs), ksplit,
                                             _ptr(bslab) if bslab is not None else None, _ptr(ws), ws.numel() * 4, stream)
        wgrad_launch.__name__ = "sgd_wgrad"
        self.prog.add(tag + ".wgrad", wgrad_launch, flops=2.0 * rows * cout * cin * taps)
        dw = dw_view if dw_view is not None else self.pg(wname)
        if bias_name is not None:
            # weight and bias gradient of the layer folded by one launch
            self.prog.add(tag + ".wred", self.lib.sgd_wgrad_reduce_bias, _ptr(slabs), ksplit, taps, cout, cin, _ptr(dw), 0,
                          self.unscale, _ptr(bslab), brows, _ptr(self.pg(bias_name)))
        else:
            self.prog.add(tag + ".wred", self.lib.sgd_wgrad_reduce, _ptr(slabs), ksplit, taps, cout, cin, _ptr(dw), 0,
                          self.unscale)
        if dw_view is None:
            self.wrote(wname)
        if bias_name is not None:
            self.wrote(bias_name)

    def _wscratch(self):
        if getattr(self, "_wscratch_buf", None) is None or self._wscratch_buf.numel() * 4 < self.wscratch_bytes:
            self._wscratch_buf = torch.empty(max(4, self.wscratch_bytes) // 4 + 4, dtype=torch.float32, device=self.dev)
        return self._wscratch_buf

    def colsum(self, tag, g_ptr, rows, c, ld, pname):
        self.prog.add(tag, self.lib.sgd_colsum, g_ptr, rows, c, ld, _ptr(self.pg(pname)), 0, self.unscale,
                      _ptr(self.cwork), self.CW)
        self.wrote(pname)

    def ln_bwd(self, tag, x, g, rows, c, gamma_name, dst, acc, beta_name=None, gres=None):
        """LayerNorm backward: dst (+)= dx (+ gres); dgamma (and dbeta when the norm has a trainable bias)"""
        gxh = self.buf(rows, c)
        self.prog.add(tag, self.lib.sgd_ln_bwd, _ptr(x), _ptr(g), _ptr(self.m.P(gamma_name)), rows, c, 1e-5, _ptr(dst),
                      int(acc), _ptr(gxh), _ptr(gres) if gres is not None else None)
        self.colsum(tag + ".dgamma", _ptr(gxh), rows, c, c, gamma_name)
        if beta_name is not None:
            self.colsum(tag + ".dbeta", _ptr(g), rows, c, c, beta_name)

    def gather_op(self, tag, pname, padded, pad):
        """parameter gradient <- the entries of a zero-padded gradient that belong to the parameter (unet._Pad.gather)"""
        dst = self.pg(pname)

        def op(stream):
            dst.copy_(pad.gather(padded.reshape(-1, padded.shape[-1])).reshape(dst.shape))
            return 0
        self.prog.add(tag, op)
        self.wrote(pname)

    def copy_op(self, tag, dst, src_view):
        def op(stream):
            dst.copy_(src_view)
            return 0
        self.prog.add(tag, op)

    def gn_bwd(self, tag, srcs, hw, a, b, sums, gname, silu, gu, gu_ld, gu_mode, gres, gres_ld, gres_mode,
               film_ptr=0, film_ld=0, dfilm_ptr=0, drop=(0.0, 0)):
        """GroupNorm(+FiLM)[+SiLU] backward over a (virtual concat of) source tensor(s); writes/accumulates the
        input gradients into the sources' gradient buffers."""
        n, lib = self.n, self.lib
        h, w = hw
        ct = sum(c for _, c in srcs)
        # Row-stream reduce (round 6, csrc/backward.hip): same-resolution gradient, every source a power-of-two channel count whose
        # image (>= 32 x 32) is a whole number of 32 KiB windows.  It writes per-chunk partial sums (one table for all sources:
        # the smallest of their chunk counts) that the coefficient launch folds itself.
        rows_ok = gu_mode == L.RS_NONE and os.environ.get("SGDM_GN_BWD_ROWS", "1") != "0"
        ks = [int(lib.sgd_gn_bwd_rows_chunks(n, h, w, c)) for _, c in srcs] if rows_ok else [0]
        chunks = min(ks) if min(ks) > 0 and all(k % min(ks) == 0 for k in ks) else 0
        S = self.buf(n, max(1, chunks), ct, 2)
        off = 0
        for t, c in srcs:
            if chunks:
                self.prog.add(tag + ".reduce", lib.sgd_gn_bwd_reduce_rows, _ptr(t), n, h, w, c, ct, off, _ptr(a), _ptr(b), silu,
                              _ptr(gu), gu_ld, drop[0], drop[1], chunks, _ptr(S), nbytes=8.0 * n * h * w * c)
            else:
                self.prog.add(tag + ".reduce", lib.sgd_gn_bwd_reduce, _ptr(t), n, h, w, c, ct, off, _ptr(a), _ptr(b), silu,
                              _ptr(gu), gu_ld, gu_mode, drop[0], drop[1], _ptr(S), nbytes=8.0 * n * h * w * c)
            off += c
        sch = max(1, chunks)
        A, B, Cc = (self.buf(n, ct) for _ in range(3))
        gamma, beta = self.m.P(gname + ".weight"), self.m.P(gname + ".bias")
        # (the launcher's bound: dynamic table + the kernel's 2 KB of static reduction scratch inside the 64 KB default limit)
        fused = (n <= 256 and (16 if chunks else 8) * n * (ct // GN_GROUPS) + 32 * n + 4 + 2048 <= 64 * 1024
                 and os.environ.get("SGDM_GN_BWD_FOLD", "1") != "0")
        if fused:                # coefficients + dgamma / dbeta column sums in ONE launch (bit-identical to the two below)
            self.prog.add(tag + ".coef", lib.sgd_gn_bwd_coef_fold, _ptr(S), sch, _ptr(sums), _ptr(gamma), _ptr(beta),
                          C.c_void_p(film_ptr), film_ld, n, ct, GN_GROUPS, h * w, GN_EPS, _ptr(A), _ptr(B), _ptr(Cc),
                          C.c_void_p(dfilm_ptr), _ptr(self.pg(gname + ".weight")), _ptr(self.pg(gname + ".bias")), 0,
                          self.unscale)
        else:
            dg, db = self.buf(n, ct), self.buf(n, ct)
            self.prog.add(tag + ".coef", lib.sgd_gn_bwd_coef, _ptr(S), sch, _ptr(sums), _ptr(gamma), _ptr(beta),
                          C.c_void_p(film_ptr), film_ld, n, ct, GN_GROUPS, h * w, GN_EPS, _ptr(A), _ptr(B), _ptr(Cc),
                          _ptr(dg), _ptr(db), C.c_void_p(dfilm_ptr))
        if fused:
            pass
        elif n <= 256:           # dgamma and dbeta from the per-sample tables in one launch
            self.prog.add(tag + ".dgamma_dbeta", lib.sgd_colsum_pair, _ptr(dg), _ptr(db), n, ct, ct,
                          _ptr(self.pg(gname + ".weight")), _ptr(self.pg(gname + ".bias")), 0, self.unscale)
        else:
            self.prog.add(tag + ".dgamma", lib.sgd_colsum, _ptr(dg), n, ct, ct, _ptr(self.pg(gname + ".weight")), 0,
                          self.unscale, _ptr(self.cwork), self.CW)
            self.prog.add(tag + ".dbeta", lib.sgd_colsum, _ptr(db), n, ct, ct, _ptr(self.pg(gname + ".bias")), 0,
                          self.unscale, _ptr(self.cwork), self.CW)
        self.wrote(gname + ".weight")
        self.wrote(gname + ".bias")
        off = 0
        for t, c in srcs:
            dst, acc = self.gact(t)
            nb = 4.0 * n * h * w * c * (3 + (1 if gres is not None else 0) + (1 if acc else 0))
            self.prog.add(tag + ".apply", lib.sgd_gn_bwd_apply, _ptr(t), n, h, w, c, ct, off, _ptr(a), _ptr(b), silu,
                          _ptr(gu), gu_ld, gu_mode, drop[0], drop[1], _ptr(A), _ptr(B), _ptr(Cc),
                          _ptr(gres) if gres is not None else None,
                          gres_ld, gres_mode, _ptr(dst), c, 0, acc, nbytes=nb)
            off += c

    # ---------------------------------------------------------------- program
    def _build(self):
        e, m, n = self.e, self.m, self.n
        P = m.P
        self.geps = self.buf(*e.eps_nhwc.shape)           # filled by the autograd Function before the program runs
        film_rec = [r for r in e.tape if r["kind"] == "film"][0]
        self.gfilm = self.buf(n, film_rec["film_w"])
        self._film_plan(film_rec)
        for rec in reversed(e.tape):
            kind = rec["kind"]
            if kind == "head":
                self._head(rec)
            elif kind == "res":
                self._res(rec)
            elif kind == "attn":
                self._attn(rec)
            elif kind == "conv_in":
                h, w = rec["hw"]
                gy = self.gread(rec["y"])
                self.wgrad(rec["p"], rec["a"], gy, rec["cout"], rec["cout"], rec["cin"], 9, n * h * w,
                           rec["p"] + ".weight", rec["p"] + ".bias")
            elif kind == "film":
                self._film(rec)
            elif kind == "mlp2":
                self._mlp2(rec)
            elif kind == "attn_lr":
                self._attn_lr(rec)
            elif kind == "down":
                self._down(rec)
            elif kind == "up":
                self._up(rec)
            elif kind == "norm_cond":
                self._norm_cond(rec)
            elif kind == "mlp_chain":
                self._mlp_chain(rec)
            elif kind == "linear":
                gy = self.gread(rec["y"])
                self.wgrad(rec["name"], rec["a"], gy, rec["cout"], rec["cout"], rec["cin"], 1, n,
                           rec["name"] + ".weight", rec["name"] + ".bias")
            else:
                raise NotImplementedError(kind)

    def _head(self, rec):
        n, (h, w), c = self.n, rec["hw"], rec["c"]
        oc = self.m.out_channels
        gu = self.buf(n, h, w, c)
        wt = self.m.P("out.2.weight")
        self.dgrad("out.2.dgrad", self.geps, oc, gu, c, [wt], lambda: wt, oc, c, 3, conv=(n, h, w))
        self.wgrad("out.2", rec["conv"], self.geps, oc, oc, c, 9, n * h * w, "out.2.weight", "out.2.bias")
        self.gn_bwd("out.0", [(rec["x"], c)], (h, w), rec["a"], rec["b"], rec["sums"], "out.0", 1, gu, c, L.RS_NONE,
                    None, 0, 0)

    def _res(self, rec):
        n, p = self.n, rec["p"]
        P = self.m.P
        cin, cout = rec["cin"], rec["cout"]
        (hh, ww), (ho, wo), rs = rec["hw_in"], rec["hw_out"], rec["rs"]
        gy = self.gread(rec["y"])
        rows_o = n * ho * wo
        # conv2 (out_layers.3)
        w2 = P(p + ".out_layers.3.weight")
        gu2 = self.buf(n, ho, wo, cout)
        self.dgrad(p + ".out_layers.3.dgrad", gy, cout, gu2, cout, [w2], lambda w2=w2: w2, cout, cout, 3, conv=(n, ho, wo))
        self.wgrad(p + ".out_layers.3", rec["conv2"], gy, cout, cout, cout, 9, rows_o, p + ".out_layers.3.weight",
                   p + ".out_layers.3.bias")
        # GN2 + FiLM + SiLU -> gradient of h1 and of this block's FiLM slice
        e = self.e
        film_ptr = e.film.data_ptr() + 4 * rec["film_off"]
        dfilm_ptr = self.gfilm.data_ptr() + 4 * rec["film_off"]
        self.G[rec["h1"].data_ptr()] = [self.buf(*rec["h1"].shape), False]
        if rec.get("ss", True):
            self.gn_bwd(p + ".out_layers.0", [(rec["h1"], cout)], (ho, wo), rec["a2"], rec["b2"], rec["sums2"],
                        p + ".out_layers.0", 1, gu2, cout, L.RS_NONE, None, 0, 0, film_ptr=film_ptr, film_ld=e.film_ld,
                        dfilm_ptr=dfilm_ptr, drop=(rec["drop_p"], rec["drop_seed"]))
            gh1 = self.gread(rec["h1"])
        else:
            # additive embedding (openaimodel.py:317-319): h1 holds h + emb_out; the gradient of emb_out[n, c] is the pixel sum of
            # the gradient of h1 -- the (sum, .) column of a statistics pass over it -- and lands in this block's slice of the
            # embedding-projection gradient like the FiLM form's (dscale, dshift)
            self.gn_bwd(p + ".out_layers.0", [(rec["h1"], cout)], (ho, wo), rec["a2"], rec["b2"], rec["sums2"],
                        p + ".out_layers.0", 1, gu2, cout, L.RS_NONE, None, 0, 0, drop=(rec["drop_p"], rec["drop_seed"]))
            gh1 = self.gread(rec["h1"])
            st = self.buf(n, cout, 2)
            self.prog.add(p + ".emb_add.bwd", self.lib.sgd_chan_stats, _ptr(gh1), n, ho * wo, cout, _ptr(st), cout, 0)
            self.copy_op(p + ".emb_add.gfilm", self.gfilm[:, rec["film_off"]:rec["film_off"] + cout], st[:, :, 0])
        self._film_group_done(p)
        # conv1 (in_layers.2)
        w1 = P(p + ".in_layers.2.weight")
        gu1 = self.buf(n, ho, wo, cin)
        self.dgrad(p + ".in_layers.2.dgrad", gh1, cout, gu1, cin, [w1], lambda w1=w1: w1, cout, cin, 3, conv=(n, ho, wo))
        self.wgrad(p + ".in_layers.2", rec["conv1"], gh1, cout, cout, cin, 9, rows_o, p + ".in_layers.2.weight",
                   p + ".in_layers.2.bias")
        # skip path
        if rec["skip"] is not None:
            ws = P(p + ".skip_connection.weight")
            gsk = self.buf(n, hh, ww, cin)
            self.dgrad(p + ".skip.dgrad", gy, cout, gsk, cin, [ws], lambda ws=ws: ws, cout, cin, 1, m=n * hh * ww)
            self.wgrad(p + ".skip_connection", rec["skip"], gy, cout, cout, cin, 1, n * hh * ww,
                       p + ".skip_connection.weight", p + ".skip_connection.bias")
            gres, gres_ld, gres_mode = gsk, cin, L.RS_NONE
        else:
            gres, gres_ld, gres_mode = gy, cout, rs            # identity skip through the same resample
        # GN1 + SiLU (+ resample adjoint of the activated tensor) -> gradients of the block inputs
        self.gn_bwd(p + ".in_layers.0", rec["srcs"], (hh, ww), rec["a1"], rec["b1"], rec["sums1"], p + ".in_layers.0",
                    1, gu1, cin, rs, gres, gres_ld, gres_mode)

    def wgrad_padded(self, tag, fwd, g, cout, cin, nrows, wname, wpad, bias_name=None, bpad=None):
        """weight (and bias) gradient of a 1x1 / linear layer whose packed operator is a zero-padded re-layout of the parameter
        (unet._Pad; cout / cin are the PADDED widths): the padded gradient is gathered back into the parameter's own shape -- the
        padding rows / columns receive gradients that belong to no parameter.  bpad None with a bias: the bias is not padded."""
        if wpad is None:
            return self.wgrad(tag, fwd, g, cout, cout, cin, 1, nrows, wname, bias_name)
        tmp = self.buf(cout, cin)
        self.wgrad(tag, fwd, g, cout, cout, cin, 1, nrows, wname, None, dw_view=tmp)
        self.gather_op(tag + ".unpad", wname, tmp, wpad)
        if bias_name is None:
            return
        if bpad is None:
            return self.colsum(tag + ".bias", _ptr(g), nrows, cout, cout, bias_name)
        tb = self.buf(1, cout)             # bias gradient = column sums of g, gathered like the weight's rows
        self.prog.add(tag + ".bias", self.lib.sgd_colsum, _ptr(g), nrows, cout, cout, _ptr(tb), 0, self.unscale,
                      _ptr(self.cwork), self.CW)
        self.gather_op(tag + ".bias.unpad", bias_name, tb, bpad)

    def _attn(self, rec):
        """AttentionBlock backward (autograd of openaimodel.py:365-371, 403-420 / 427-451); zero-padded heads (rec["pads"]) as in
        _attn_lr: every tensor between qkv and proj_out is dp wide per head"""
        n, p, ch, heads, d, T = self.n, rec["p"], rec["ch"], rec["heads"], rec["d"], rec["T"]
        dp, pads = rec.get("dp", d), rec.get("pads")
        inner = heads * dp
        P = self.m.P
        hh, ww = rec["hw"]
        gy = self.gread(rec["y"])
        adj = lambda w, key: (lambda: w) if pads is None else (lambda: pads[key].apply(w.detach().float()))
        pad = lambda key: None if pads is None else pads[key]
        wp = P(p + ".proj_out.weight")
        gatt = self.buf(n, T, inner)
        self.dgrad(p + ".proj_out.dgrad", gy, ch, gatt, inner, [wp], adj(wp, "out"), ch, inner, 1, m=n * T)
        self.wgrad_padded(p + ".proj_out", rec["proj_args"], gy, ch, inner, n * T, p + ".proj_out.weight", pad("out"),
                          p + ".proj_out.bias")
        qkv, gqkv = rec["qkv"], self.buf(n, T, 3 * inner)
        dvec = self.buf(n, heads, T)
        off = lambda t, k: C.c_void_p(t.data_ptr() + 4 * k)
        hs, ko, vo = rec.get("qkv_layout", (3 * d, d, 2 * d))      # head stride, k / v offsets: legacy or new attention order
        self.prog.add(p + ".attn_bwd", self.attention_bwd_fn(dp), _ptr(qkv), 3 * inner, hs, off(qkv, ko), off(qkv, vo),
                      3 * inner, hs, _ptr(rec["att"]), inner, _ptr(gatt), inner, _ptr(rec["lse"]), _ptr(dvec), n, heads, T, T,
                      dp, 1.0 / math.sqrt(d), _ptr(gqkv), off(gqkv, ko), off(gqkv, vo))
        wq = P(p + ".qkv.weight")
        gxn = self.buf(n, T, ch)
        self.dgrad(p + ".qkv.dgrad", gqkv, 3 * inner, gxn, ch, [wq], adj(wq, "qkv"), 3 * inner, ch, 1, m=n * T)
        self.wgrad_padded(p + ".qkv", rec["qkv_args"], gqkv, 3 * inner, ch, n * T, p + ".qkv.weight", pad("qkv"),
                          p + ".qkv.bias", pad("qkv_bias"))
        self.gn_bwd(p + ".norm", [(rec["x"], ch)], (hh, ww), rec["a"], rec["b"], rec["sums"], p + ".norm", 0, gxn, ch,
                    L.RS_NONE, gy, ch, L.RS_NONE)                       # residual: x + proj(...)

    def _down(self, rec):
        """Downsample = conv3x3 stride 2 (openaimodel_ca.py:167-174)"""
        n, p, c = self.n, rec["p"], rec["c"]
        hh, ww = rec["hw_in"]
        gy = self.gread(rec["y"])
        w = self.m.P(p + ".op.weight")
        dst, acc = self.gact(rec["x"])
        self.dgrad(p + ".op.dgrad", gy, c, dst, c, [w], lambda: w, c, c, 3, conv=(n, hh // 2, ww // 2), zero_up=True,
                   acc=bool(acc))
        self.wgrad(p + ".op", rec["a"], gy, c, c, c, 9, n * (hh // 2) * (ww // 2), p + ".op.weight", p + ".op.bias")

    def _up(self, rec):
        """Upsample = nearest x2 + conv3x3 (openaimodel_ca.py:128-131)"""
        n, p, c = self.n, rec["p"], rec["c"]
        hh, ww = rec["hw_in"]
        gy = self.gread(rec["y"])
        w = self.m.P(p + ".conv.weight")
        gu = self.buf(n, 2 * hh, 2 * ww, c)
        self.dgrad(p + ".conv.dgrad", gy, c, gu, c, [w], lambda: w, c, c, 3, conv=(n, 2 * hh, 2 * ww))
        dst, acc = self.gact(rec["x"])
        self.prog.add(p + ".up_adj", self.lib.sgd_resample_bwd, _ptr(gu), n, hh, ww, c, L.RS_UP2, _ptr(dst), acc)
        self.wgrad(p + ".conv", rec["a"], gy, c, c, c, 9, n * 4 * hh * ww, p + ".conv.weight", p + ".conv.bias")

    def _attn_lr(self, rec):
        """Attention_LR backward (autograd of crossattetion_lr.py:81-142).  With zero-padded heads (rec["pads"], head widths
        the attention core has no instance for) every tensor between the projections is dp wide per head, the adjoint
        operators are built from the padded weights and each padded weight gradient is gathered back into the parameter's
        own shape (the padding rows / columns receive gradients that belong to no parameter)."""
        n, p, ch, heads, d, T, J, ntok = (self.n, rec["p"], rec["ch"], rec["heads"], rec["d"], rec["T"], rec["J"],
                                          rec["ntok"])
        dp, pads = rec.get("dp", d), rec.get("pads")
        P, lib = self.m.P, self.lib
        x, gy = rec["x"], self.gread(rec["y"])
        rows = n * T
        inner = heads * dp

        def adj(w, key):                       # forward weight as the adjoint operator sees it
            return (lambda: w) if pads is None else (lambda: pads[key].apply(w.detach().float()))

        def wgrad(tag, fwd, g, cout, cin, nrows, wname, key, bias_name=None):
            self.wgrad_padded(tag, fwd, g, cout, cin, nrows, wname, None if pads is None else pads[key], bias_name,
                              None if pads is None else pads["vec"])

        # y = x + LN_out(o):  LN_out backward (gamma trainable, beta is a buffer)
        go = self.buf(n, T, ch)
        self.ln_bwd(p + ".to_out.1", rec["o"], gy, rows, ch, p + ".to_out.1.gamma", go, 0)
        wo = P(p + ".to_out.0.weight")
        gatt = self.buf(n, T, inner)
        self.dgrad(p + ".to_out.0.dgrad", go, ch, gatt, inner, [wo], adj(wo, "out"), ch, inner, 1, m=rows)
        wgrad(p + ".to_out.0", rec["aout"], go, ch, inner, rows, p + ".to_out.0.weight", "out")
        # multi-query attention core
        q, kv = rec["q"], rec["kv"]
        gq, gkv = self.buf(n, T, inner), self.buf(n, J, 2 * dp)
        dvec = self.buf(n, heads, T)
        self.prog.add(p + ".attn_bwd", self.attention_bwd_fn(dp), _ptr(q), inner, dp, _ptr(kv),
                      C.c_void_p(kv.data_ptr() + 4 * dp), 2 * dp, 0, _ptr(rec["att"]), inner, _ptr(gatt), inner,
                      _ptr(rec["lse"]), _ptr(dvec), n, heads, T, J, dp, d ** -0.5, _ptr(gq), _ptr(gkv),
                      C.c_void_p(gkv.data_ptr() + 4 * dp))
        # to_q / to_kv share LN(x): gradient of the normalised input is the sum of both adjoints
        wq, wkv = P(p + ".to_q.weight"), P(p + ".to_kv.weight")
        gxn = self.buf(n, T, ch)
        self.dgrad(p + ".to_q.dgrad", gq, inner, gxn, ch, [wq], adj(wq, "q"), inner, ch, 1, m=rows)
        wgrad(p + ".to_q", rec["aq"], gq, inner, ch, rows, p + ".to_q.weight", "q")
        gkv_self = self.buf(n, T, 2 * dp)
        self.copy_op(p + ".gkv_self", gkv_self, gkv[:, ntok + 1:, :])
        self.dgrad(p + ".to_kv.dgrad", gkv_self, 2 * dp, gxn, ch, [wkv], adj(wkv, "kv"), 2 * dp, ch, 1, m=rows, acc=True)
        wgrad(p + ".to_kv", rec["akv"], gkv_self, 2 * dp, ch, rows, p + ".to_kv.weight", "kv")
        # null key/value: summed over the batch
        if pads is None:
            self.colsum(p + ".null_kv", C.c_void_p(gkv.data_ptr() + 4 * ntok * 2 * d), n, 2 * d, J * 2 * d, p + ".null_kv")
        else:
            tn = self.buf(2, dp)
            self.prog.add(p + ".null_kv", lib.sgd_colsum, C.c_void_p(gkv.data_ptr() + 4 * ntok * 2 * dp), n, 2 * dp,
                          J * 2 * dp, _ptr(tn), 0, self.unscale, _ptr(self.cwork), self.CW)
            self.gather_op(p + ".null_kv.unpad", p + ".null_kv", tn, pads["null"])
        # context keys/values -> to_context.1 (Linear) -> to_context.0 (LayerNorm) -> shared context tokens
        ctx = rec["ctx"]
        gckv = self.buf(n, ntok, 2 * dp)
        self.copy_op(p + ".gkv_ctx", gckv, gkv[:, :ntok, :])
        wc = P(p + ".to_context.1.weight")
        wgrad(p + ".to_context.1", rec["actx"], gckv, 2 * dp, ctx, n * ntok, p + ".to_context.1.weight", "kv",
              p + ".to_context.1.bias")
        gcn = self.buf(n, ntok, ctx)
        self.dgrad(p + ".to_context.1.dgrad", gckv, 2 * dp, gcn, ctx, [wc], adj(wc, "kv"), 2 * dp, ctx, 1, m=n * ntok)
        cdst, cacc = self.gact(rec["context"])
        self.ln_bwd(p + ".to_context.0", rec["context"], gcn, n * ntok, ctx, p + ".to_context.0.weight", cdst, cacc,
                    beta_name=p + ".to_context.0.bias")
        # input LayerNorm (gamma trainable, beta buffer) + the residual path
        dst, acc = self.gact(x)
        self.ln_bwd(p + ".norm", x, gxn, rows, ch, p + ".norm.gamma", dst, acc, gres=gy)

    def _norm_cond(self, rec):
        """context = LayerNorm(cat(time tokens, cond tokens)) (openaimodel_ca.py:973,1017)"""
        n, ntok, ctx, wt = self.n, rec["ntok"], rec["ctx"], rec["wt"]
        gctx = self.gread(rec["context"])
        graw = self.buf(n, ntok, ctx)
        self.ln_bwd("norm_cond", rec["raw"], gctx, n * ntok, ctx, "norm_cond.weight", graw, 0, beta_name="norm_cond.bias")
        g2 = graw.view(n, ntok * ctx)
        gt = self.buf(n, wt)
        self.copy_op("norm_cond.split_t", gt, g2[:, :wt])
        self.G[rec["raw_t"].data_ptr()] = [gt, True]
        if rec["raw_c"] is not None:
            gc = self.buf(n, ntok * ctx - wt)
            self.copy_op("norm_cond.split_c", gc, g2[:, wt:])
            self.G[rec["raw_c"].data_ptr()] = [gc, True]

    def _film(self, rec):
        """emb_layers of all ResBlocks (one GEMM forward): weight/bias grads split back per block; gradient of
        SiLU(emb) through the adjoint, then through the SiLU, into the (virtual concat) emb parts."""
        n, m = self.n, self.m
        fw = rec["film_w"]
        ted = rec["emb_t"].shape[1]
        cc = rec["emb_c"].shape[1] if rec["emb_c"] is not None else 0
        ech = ted + cc
        names = rec["names"]
        wparams = [m.P(p + ".emb_layers.1.weight") for p in names]
        # (the weight / bias gradients of the FiLM projections were produced stage by stage inside the walk: _film_group_done)
        assert not self._film_open, self._film_open
        cat = lambda: torch.cat([w.detach() for w in wparams], 0)
        parts = [(rec["emb_t"], 0, ted)] + ([(rec["emb_c"], ted, cc)] if cc else [])
        # few rows (the batch), a very long reduction (all FiLM outputs, 13,824 at C2): split-K on the concatenated weight as
        # stored -- the conv kernel walked it with 4 tiles (0.75 ms per step); the concatenation follows the parameters
        skinny = n <= 256 and fw >= 2048 and os.environ.get("SGDM_SKINNY_DGRAD", "1") != "0"
        if skinny:
            wcat = self.buf(fw, ech)
            box = dict(sig=None)

            def refresh_wcat(stream, wcat=wcat, box=box):
                sig = tuple((w.data_ptr(), w._version) for w in wparams)
                if sig != box["sig"]:
                    torch.cat([w.detach().float() for w in wparams], 0, out=wcat)
                    box["sig"] = sig
                return 0
            self.prog.add("emb_layers.wcat", refresh_wcat)
            # y[n, ech] = sum_k gfilm[n, k] wcat[k, ech] IS a weight gradient seen from the other side: "rows" = the fw FiLM
            # outputs, "gy" = gfilm transposed (fw x n), "a" = wcat.  The split 1x1 weight-gradient kernel (MFMA, split K over
            # the rows) replaces the fp32-FMA sgd_linear_splitk_t (LDS-bound: 0.25 ms per step at C2 for 1.7 GFLOP)
            gfT, gall = self.buf(fw, n), self.buf(n, ech)
            self.copy_op("emb_layers.gfilm_t", gfT, self.gfilm.view(n, fw).t())
            wa = L.IgemmArgs()
            wa.x0, wa.c0, wa.mode, wa.m, wa.stride, wa.prec = wcat.data_ptr(), ech, L.MODE_FLAT, fw, 1, self.prec
            self.keep.append(wa)
            ksplit = max(1, min((fw + 63) // 64 // 4, 512 // ((ech + 127) // 128)))
            slabs = self.buf(ksplit, 1, n, ech)
            lib = self.lib

            def sgd_wgrad(stream):
                ws = self._wscratch()
                return lib.sgd_wgrad_scratch(C.byref(wa), _ptr(gfT), n, n, _ptr(slabs), ksplit, None, _ptr(ws), ws.numel() * 4,
                                             stream)
            self.prog.add("emb_layers.dgrad", sgd_wgrad, flops=2.0 * n * ech * fw)
            self.prog.add("emb_layers.dgrad.fold", lib.sgd_wgrad_reduce, _ptr(slabs), ksplit, 1, n, ech, _ptr(gall), 0, 1.0)
        for t, o, c in parts:
            gact = self.buf(n, c)
            if skinny:
                self.copy_op(f"emb_layers.dgrad.part{o}", gact, gall[:, o:o + c])
            else:
                self.dgrad(f"emb_layers.dgrad{o}", self.gfilm, fw, gact, c, wparams, lambda o=o, c=c: cat()[:, o:o + c], fw, c,
                           1, m=n)
            g = self.buf(n, c)
            self.prog.add(f"emb.silu_bwd{o}", self.lib.sgd_silu_bwd, _ptr(t), _ptr(gact), n * c, _ptr(g))
            self.G[t.data_ptr()] = [g, True]

    # FiLM projections (emb_layers of all ResBlocks: ONE GEMM in the forward, its 42 MB of weight gradients the largest single
    # tensor group of the model).  As one launch at the end of the walk -- its gradient rows are complete only when the LAST
    # ResBlock's GroupNorm backward has written its slice -- it was the last gradient of the backward: on 8 GPUs the part of the
    # exchange nothing hides (round 5: bucket 4 + tail, ~80 MB, complete 0.4 ms before the end).  The walk visits the
    # ResBlocks in reverse forward order and a block's slice of the FiLM gradient is final after its out_layers.0 backward,
    # so the columns are cut into STAGES of consecutive blocks (>= FILM_STAGE_BYTES of weight gradient each) and every
    # stage's weight / bias gradient is launched as soon as its first (in forward order) block is through.
    FILM_STAGE_BYTES = 6 << 20

    def _film_plan(self, rec):
        names, couts = rec["names"], rec["couts"]
        fw = rec["film_w"]
        ech = rec["emb_t"].shape[1] + (rec["emb_c"].shape[1] if rec["emb_c"] is not None else 0)
        offs, off = [], 0
        fm = self.m._film_mult                     # 2: FiLM (scale, shift); 1: additive embedding (use_scale_shift_norm=False)
        for co in couts:
            offs.append(off)
            off += fm * co
        self._film_ctx = dict(rec=rec, ech=ech, fw=fw, dwcat=self.buf(fw, ech), dbcat=self.buf(fw))
        # stages in WALK order (last ResBlock of the forward first); a stage closes at block i when it holds enough bytes
        self._film_close = {}                       # block name -> (first block index, one past the last) of the stage it closes
        hi = len(names)
        acc = 0
        for i in range(len(names) - 1, -1, -1):
            acc += fm * couts[i] * ech * 4
            if acc >= self.FILM_STAGE_BYTES or i == 0:
                self._film_close[names[i]] = (i, hi)
                hi, acc = i, 0
        self._film_offs = offs + [fw]
        self._film_open = set(self._film_close)

    def _film_group_done(self, p):
        """called from the walk right after ResBlock `p`'s out_layers.0 backward has written its FiLM-gradient slice"""
        grp = self._film_close.get(p)
        if grp is None:
            return
        self._film_open.discard(p)
        i0, i1 = grp
        f, n = self._film_ctx, self.n
        rec, ech, fw, dwcat, dbcat = f["rec"], f["ech"], f["fw"], f["dwcat"], f["dbcat"]
        names, couts = rec["names"], rec["couts"]
        c0, c1 = self._film_offs[i0], self._film_offs[i1]
        tag = f"emb_layers.s{i0}"
        gsl = self.gfilm[:, c0:c1]                  # columns of this stage (row stride fw)
        dw_dst, db_dst = dwcat[c0:c1], dbcat[c0:c1]
        if self.arena is None:
            # single process: the per-block gradients ARE row slices of the concatenated result (contiguous), no copies.
            # (first-use order = the arena's order in a data-parallel build: the stage's weights, then its biases)
            for i in range(i0, i1):
                self.pgrad[names[i] + ".emb_layers.1.weight"] = dwcat[self._film_offs[i]:self._film_offs[i + 1]]
            for i in range(i0, i1):
                self.pgrad[names[i] + ".emb_layers.1.bias"] = dbcat[self._film_offs[i]:self._film_offs[i + 1]]
        else:
            # data parallel: the stage's weights, then its biases, take CONSECUTIVE arena slots (first-use order; every slot a
            # whole number of quads), so the stage's rows of the concatenated gradient ARE one arena range each: the reduce and
            # the column sums write there directly (round 6: 44 device-side slice copies per step fewer)
            wv = [self.pg(names[i] + ".emb_layers.1.weight") for i in range(i0, i1)]
            bv = [self.pg(names[i] + ".emb_layers.1.bias") for i in range(i0, i1)]
            flat = self.arena.flat

            def span(views, rows, cols):
                o0 = (views[0].data_ptr() - flat.data_ptr()) // flat.element_size()
                t = flat[o0:o0 + rows * cols].view(rows, cols) if cols > 1 else flat[o0:o0 + rows]
                off = 0
                for v in views:                     # consecutive, unpadded: each view starts where the one before it ended
                    assert v.data_ptr() == t.data_ptr() + off * flat.element_size(), "FiLM stage is not one arena range"
                    off += v.numel()
                return t
            dw_dst, db_dst = span(wv, c1 - c0, ech), span(bv, c1 - c0, 1)
        self.wgrad(tag, rec["a"], gsl, fw, c1 - c0, ech, 1, n, None, None, dw_view=dw_dst)
        self.prog.add(tag + ".bias", self.lib.sgd_colsum, _ptr(gsl), n, c1 - c0, fw, _ptr(db_dst), 0, self.unscale,
                      _ptr(self.cwork), self.CW)
        for i in range(i1 - 1, i0 - 1, -1):
            self.wrote(names[i] + ".emb_layers.1.weight")
            self.wrote(names[i] + ".emb_layers.1.bias")

    def _mlp2(self, rec):
        """Linear -> SiLU -> Linear (time_embed / mlp_cond)"""
        n, name = self.n, rec["name"]
        P = self.m.P
        gy = self.gread(rec["y"])
        cin, mid, cout = rec["cin"], rec["mid"], rec["cout"]
        self.wgrad(name + ".2", rec["a2"], gy, cout, cout, mid, 1, n, name + ".2.weight", name + ".2.bias")
        w2 = P(name + ".2.weight")
        gh_act = self.buf(n, mid)
        self.dgrad(name + ".2.dgrad", gy, cout, gh_act, mid, [w2], lambda: w2, cout, mid, 1, m=n)
        gh = self.buf(n, mid)
        self.prog.add(name + ".silu_bwd", self.lib.sgd_silu_bwd, _ptr(rec["h"]), _ptr(gh_act), n * mid, _ptr(gh))
        self.wgrad(name + ".0", rec["a0"], gh, mid, mid, cin, 1, n, name + ".0.weight", name + ".0.bias")

    def _mlp_chain(self, rec):
        """to_cond_tokens_2d (openaimodel_ca.py:606-614): Linear -> SiLU -> Linear -> SiLU -> Linear -> SiLU -> Linear over
        the n*T token rows; every SiLU was the next GEMM's prologue, so each layer's input tensor is pre-activation"""
        rows, P = rec["rows"], self.m.P
        g = self.gread(rec["y"])                          # [n, T*ctx] == [n*T, ctx] rows of the last layer's output
        layers = rec["layers"]
        for li in range(len(layers) - 1, -1, -1):
            ly = layers[li]
            name, cin, cout = ly["name"], ly["cin"], ly["cout"]
            self.wgrad(name, ly["a"], g, cout, cout, cin, 1, rows, name + ".weight", name + ".bias")
            if li == 0:
                break                                      # the layer input is the guidance itself: no gradient wanted
            w = P(name + ".weight")
            gact = self.buf(rows, cin)
            self.dgrad(name + ".dgrad", g, cout, gact, cin, [w], lambda w=w: w, cout, cin, 1, m=rows)
            gpre = self.buf(rows, cin)
            self.prog.add(name + ".silu_bwd", self.lib.sgd_silu_bwd, _ptr(ly["x"]), _ptr(gact), rows * cin, _ptr(gpre))
            g = gpre

    # ---------------------------------------------------------------- execution
    def run(self, geps_nchw):
        lib = self.lib
        stream = torch.cuda.current_stream().cuda_stream
        if getattr(self, "_pack_batch", None) is None or len(self._pack_batch.packs) != len(self.packs):
            from .unet import _PackBatch
            self._pack_batch = _PackBatch(self.packs, self.prec, self.dev)
        self._pack_batch.refresh(stream)
        for a, pk in self.late:
            a.cin_p, a.cout_p = pk.cin_p, pk.cout_p
        n, c = self.n, self.m.out_channels
        h, w = self.e.h, self.e.w
        g = (geps_nchw.float() * self.gscale).contiguous()
        L.check(lib.sgd_pack_input(_ptr(g), None, None, None, n, n, c, 0, h, w, _ptr(self.geps), stream), "geps")
        if self.reducer is not None:
            self.reducer.start()
        self.prog.run(stream)
        # Health of this step's gradients: 1.0 iff the engine's balanced-tail health word is up.  With an arena
        # the flag is the arena's last slot and travels with the LAST bucket (SUM over the ranks: every rank sees > 0 when any
        # rank failed); it is added to the device's sticky gate, which the fused optimizer takes as skip_if_nonzero -- a
        # poisoned step is never applied -- and copied to pinned host memory for the next step's poll_health() to raise on.
        word, flag = self.e.health_word(), None
        if word is not None:
            from .unet import grad_health
            flag = self.arena.health if self.arena is not None else self._own_flag()
            flag.copy_(word.ne(0))
        if self.reducer is not None:
            self.reducer.backward_done()     # (overlap record: everything the exchange could hide behind is issued)
            self.reducer.finish()            # flush the tail bucket (it holds the flag), join the side stream
        if flag is not None:
            grad_health(self.dev).add_(flag)
        self.e.note_health(flag)
        return self.pgrad

    # ---- CU reserve only while a collective is in flight.  The persistent conv grid leaves
    # `reserve` compute units to RCCL's kernels on the side stream (tests/test_hip_contention.py: a launch whose blocks do
    # not all fit next to them takes up to 1.7x).  Taken from every launch of the backward the reserve cost ~1.1 ms of a
    # 63 ms step; RCCL needs it only between a bucket's enqueue and its completion.  The host knows where in the PROGRAM each
    # collective starts (the bucket hooks) and how many bytes it moves: a launch keeps the whole device unless it falls into
    # the window [hook, hook + bytes / RESERVE_GBPS + 0.2 ms) of estimated launch time (algorithmic flops at 300 TF/s, 40 us
    # for the memory-bound launches in between).  RESERVE_GBPS = 50 GB/s of all-reduced bytes (SGDM_RESERVE_GBPS) is about
    # half of what an 8-GPU xGMI ring delivers on 16 channels: a 64 MB bucket holds its window for 1.5 ms.  A STATIC rule on
    # purpose: windows sized by measured completion times would make the grid of a launch -- and with it the K split of its
    # balanced tail, i.e. the rounding of its sums -- depend on timing; this way a given program always runs the same grids.
    RESERVE_GBPS = 50.0

    def apply_grid_cap(self):
        cap = getattr(self.e, "_grid_cap", 0)
        if cap <= 0 or self.reducer is None or not self.reducer.active or os.environ.get("SGDM_RESERVE_WINDOWS", "1") == "0":
            for a, _ in self.late:
                a.grid_cap = cap
            self._windows = None
            return
        gbps = float(os.environ.get("SGDM_RESERVE_GBPS", self.RESERVE_GBPS))
        esz = self.arena.flat.element_size()
        hooks = [(i, int(op[0][len("bucket"):].split(".")[0])) for i, op in enumerate(self.prog.ops)
                 if op[0].startswith("bucket") and op[0].endswith(".allreduce")]
        est = [(mt[1] / 300e12 * 1e3 if mt[1] > 0 else 0.04) for mt in self.prog.meta]       # ms per program entry
        inside = [False] * len(self.prog.ops)
        for pos, bi in hooks:
            s0, e0, _ = self.arena.buckets[bi]
            budget = (e0 - s0) * esz / (gbps * 1e9) * 1e3 + 0.2
            j = pos
            while j < len(inside) and budget > 0:
                inside[j] = True
                budget -= est[j]
                j += 1
        for (a, _), at in zip(self.late, self.late_at):
            a.grid_cap = cap if inside[at] else 0
        self._windows = (sum(inside), len(inside))

    def _own_flag(self):
        if getattr(self, "_flag", None) is None:
            self._flag = torch.zeros(1, dtype=torch.float32, device=self.dev)
        return self._flag


class _UNetTrainFn(torch.autograd.Function):
    """eps = UNet(x, t, ...) with gradients for the trainable parameters"""

    @staticmethod
    def forward(ctx, model, eng, args, *params):
        x, t, cond, layout, mask = args
        eng.run(x, t, cond, layout, mask, train=True)
        eng.generation = getattr(eng, "generation", 0) + 1
        ctx.model, ctx.eng, ctx.generation = model, eng, eng.generation
        ctx.names = [n for n, p in model.named_parameters() if p.requires_grad]
        ctx.params = params
        return model._to_nchw(eng)

    @staticmethod
    def backward(ctx, geps):
        eng = ctx.eng
        if eng.generation != ctx.generation:
            raise RuntimeError("sgdm_amd: backward through a stale forward -- another training forward ran on this "
                               "model (same batch/resolution) after the one being differentiated and overwrote its "
                               "activations; call backward() before the next forward")
        model = ctx.model
        _check_torch_ddp(model)
        if getattr(eng, "backward", None) is None:
            eng.backward = make_backward(eng)
            eng.backward.apply_grid_cap()        # the CU reserve of a data-parallel step, inside its windows only
        # The program writes every parameter gradient into a persistent buffer (a view of the DDP arena when there is one).
        # Handing those to autograd made AccumulateGrad clone each of them -- ~390 device copies per step (it cannot steal a
        # tensor somebody else still references).  A parameter whose .grad is None gets the buffer itself as .grad (what
        # DDP's gradient_as_bucket_view does); one whose .grad already IS the buffer from an earlier backward (zero_grad(
        # set_to_none=False), gradient accumulation) is un-aliased first, so accumulation keeps torch's semantics.
        held = eng.backward.pgrad
        for name, p in zip(ctx.names, ctx.params):
            g = held.get(name)
            if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                p.grad = p.grad.clone()
        grads = eng.backward.run(geps)
        out = []
        with torch.no_grad():
            for name, p in zip(ctx.names, ctx.params):
                g = grads.get(name)          # None: parameter not on the path (e.g. to_cond_tokens_2d, README.md:90-94)
                # alias fast path only when nobody observes the gradient through autograd: tensor hooks / post-accumulate
                # hooks (gradient clipping callbacks, optimizer-in-backward, torch DDP's reducer) need AccumulateGrad to run
                alias = getattr(model, "hip_grad_alias", True) and not p._backward_hooks \
                    and not getattr(p, "_post_accumulate_grad_hooks", None)
                if g is not None and alias and p.grad is None and g.shape == p.shape:
                    p.grad = g
                    g = None
                out.append(g)
        return (None, None, None) + tuple(out)


def _check_torch_ddp(model):
    """once per model: the reference's own multi-GPU command wraps the LightningModule in torch DDP
    (pl.trainer.strategy=ddp): then torch's reducer owns the exchange -- this path must hand it the gradients through
    autograd (its hooks sit on the AccumulateGrad nodes) and must NOT all-reduce them a second time, nor broadcast the
    parameters the wrapper's constructor has already broadcast"""
    if getattr(model, "_torch_ddp_checked", False):
        return
    from .ddp import find_torch_ddp_wrapper, torch_ddp_ignores
    model._torch_ddp_checked = True
    wrapper = find_torch_ddp_wrapper(model)
    if wrapper is not None and not torch_ddp_ignores(wrapper, model):
        import warnings
        model.hip_ddp = False
        model.hip_grad_alias = False
        warnings.warn("sgdm_amd: the model is wrapped in torch DistributedDataParallel -- gradients are handed to "
                      "torch's reducer (one all-reduce, torch's buckets) and the HIP path's own overlapped RCCL "
                      "exchange is off.  For the native exchange run one process per GPU WITHOUT the wrapper "
                      "(sgdm_amd.pl_strategy.HipDDPStrategy) or call sgdm_amd.ddp.exclude_from_torch_ddp(root, "
                      "unet, ema) before wrapping (INTEGRATION.md).")


def make_backward(eng):
    """single process: plain gradient buffers.  torch.distributed initialised with world > 1: a dry build learns the
    order in which the backward produces the parameter gradients, then the real program writes them into a flat
    arena in that order and overlaps the bucketed RCCL all-reduce with the remaining launches."""
    from .ddp import BucketReducer, GradArena, exchange_active, exchange_forced, exchange_group
    if not exchange_active(eng.m):
        return Backward(eng)
    dry = Backward(eng)
    order = [(name, tuple(g.shape)) for name, g in dry.pgrad.items()]
    del dry
    arena = GradArena(order, eng.dev, bucket_bytes=getattr(eng.m, "hip_bucket_bytes", 64 << 20))
    # RCCL: a communicator of the exchange's own, capped at the CU reserve (collective over all ranks: every rank builds
    # its backward program at its first training step)
    return Backward(eng, arena, BucketReducer(arena, group=exchange_group(), average=False, force=exchange_forced(eng.m)))


def forward_train(model, x, t, cond, layout, mask, n):
    """autograd-capable UNet evaluation (called from UNetModelBase._run when grads are required)"""
    B, cx, H, W = x.shape
    if getattr(model, "use_spatial_transformer", False):
        # the reference cannot train this variant either: BasicTransformerBlock checkpoints _forward(x, context) with
        # context=None (openaimodel.py:915, attention.py:213-214) and CheckpointFunction.backward calls None.detach()
        # (diffusionmodules/util.py:132 -> AttributeError); tests/golden/make_golden_train_tokens.py documents the run
        raise NotImplementedError("training through the SpatialTransformer path is not built: the reference raises "
                                  "AttributeError in its backward for this variant (checkpoint with context=None)")
    prec = L.PREC_BY_NAME[model.hip_precision]
    if not getattr(model, "_hip_ddp_synced", False) and getattr(model, "hip_ddp", True):
        # torch DDP broadcasts rank 0's parameters and buffers when it wraps a module; nothing else on this path would, and
        # replicas that start from different weights (rank-dependent init, a missing seed_everything, per-rank
        # checkpoints) would train silently diverged: once per model, before its first exchanged training step.  Every
        # rank enters its first training forward together (it is a collective).  Under torch's own wrapper
        # (strategy=ddp unchanged) the wrapper's constructor has done this already: look for it FIRST.  A LitEma built
        # before this point holds the old values: sync the root module early instead -- HipDDPStrategy and bench.py do;
        # this late call warns on a rank whose values it had to change.
        from .ddp import exchange_active, sync_initial_state
        if exchange_active(model):
            _check_torch_ddp(model)
        if getattr(model, "hip_ddp", True):
            sync_initial_state(model, late=True)
    eng = model._engine(n, H, W, prec)
    params = [p for p in model.parameters() if p.requires_grad]
    eps = _UNetTrainFn.apply(model, eng, (x, t, cond, layout, mask), *params)
    return eps, 0.0, dict()


class _MSEFn(torch.autograd.Function):
    """per-sample mean squared error between noise and eps (ddpm.py:67-75), fused forward + gradient"""

    @staticmethod
    def forward(ctx, eps, target):
        B, Cc, H, W = eps.shape
        per = ((target - eps) ** 2).reshape(B, -1).mean(1)
        ctx.save_for_backward(eps, target)
        return per

    @staticmethod
    def backward(ctx, gper):
        eps, target = ctx.saved_tensors
        B = eps.shape[0]
        chw = eps[0].numel()
        return (-2.0 / chw) * (target - eps) * gper.reshape(B, 1, 1, 1), None


def p_losses_hip(diff, x_start, t, noise=None, *args, **kwargs):
    """LatentDiffusion.p_losses (ddpm.py:54-86)"""
    lib = L.load()
    h = diff.hparams
    noise = torch.randn_like(x_start) if noise is None else noise
    s = diff.sampler
    if x_start.device.type == "cuda":
        x_noisy = torch.empty_like(x_start)
        B = x_start.shape[0]
        x0, nz, tt = x_start.contiguous().float(), noise.contiguous().float(), t.contiguous().to(torch.int64)
        L.check(lib.sgd_q_sample(_ptr(x0), _ptr(nz), _ptr(tt), _ptr(s.sqrt_alphas_cumprod),
                                 _ptr(s.sqrt_one_minus_alphas_cumprod), B, x0[0].numel(), _ptr(x_noisy),
                                 torch.cuda.current_stream().cuda_stream), "sgd_q_sample")
    else:
        x_noisy = s.q_sample(original_sample=x_start, t=t, noise=noise)
    model_output, loss_inside, dict_inside = diff.denoise_fn(x_noisy, t, *args, **kwargs)
    prefix = "train" if diff.training else "val"
    loss_dict = {f"{prefix}/{k}": v for k, v in dict_inside.items()}
    if h.parameterization == "x0":
        target = x_start
    elif h.parameterization == "eps":
        target = noise
    else:
        raise NotImplementedError()
    if h.loss_type == "l2":
        loss = _MSEFn.apply(model_output, target)
    elif h.loss_type == "l1":
        loss = (target - model_output).abs().reshape(len(target), -1).mean(1)
    elif h.loss_type == "huber":                                              # ddpm.py:99-103
        loss = torch.nn.functional.smooth_l1_loss(target, model_output, reduction="none").reshape(len(target), -1).mean(1)
    else:
        raise NotImplementedError(f"unknown loss type '{h.loss_type}'")
    if prefix == "train":
        loss_dict[f"{prefix}/epoch_stats_y"] = loss.detach()
        loss_dict[f"{prefix}/epoch_stats_x"] = t.detach()
    loss = loss.mean()
    loss_dict[f"{prefix}/ddpm_loss"] = loss.detach()
    loss = loss + loss_inside
    loss_dict[f"{prefix}/loss"] = loss.detach()
    return loss, loss_dict
