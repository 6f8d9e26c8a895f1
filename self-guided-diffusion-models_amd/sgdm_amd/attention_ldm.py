"""``dynamic.attention_ldm.CrossAttention`` / ``LinearCrossAttention`` (reference dynamic/attention_ldm.py:198-298,
SURVEY row A23) on the HIP kernels of the hot path.

Imagen-style per-head cross-attention with a learned null key/value in front of the context and an optional context
mask; the linear variant replaces softmax(q k^T) v by softmax_d(q) (softmax_keys(k)^T v).  Same constructor keywords,
parameter / buffer names and shapes (``norm.gamma``, ``norm.beta`` [buffer], ``norm_context.*``, ``null_kv``,
``to_q.weight``, ``to_kv.weight``, ``to_out.0.weight``, ``to_out.1.gamma`` / ``.beta``) and the same forward contract as
the reference classes, so state dicts interchange.  No shipped config instantiates them (only ``log`` is imported from
that module): LayerNorm statistics + LN-prologue projections (``sgd_ln_stats`` / ``sgd_igemm``), the MFMA attention core
(``sgd_attention_split`` / ``sgd_attention`` / ``sgd_attention_masked``) or the linear core (``sgd_linear_attention``),
output projection and ``sgd_ln_apply``.  There is no CPU fallback.

The reference classes train through torch.autograd; here a grad-mode call runs as one ``torch.autograd.Function`` whose
backward is the adjoint launch sequence on the same library (``_CrossAttnFn``): ``sgd_ln_bwd`` + ``sgd_colsum`` for the three
LayerNorms, the forward implicit-GEMM kernel on adjoint-packed weights for the input gradients of the three projections,
``sgd_wgrad`` + ``sgd_wgrad_reduce`` for their weight gradients (the LayerNorm-row prologue recomputed from the raw input),
``sgd_attention_bwd`` / ``sgd_attention_masked_bwd`` / ``sgd_linear_attention_bwd`` for the core, column sums over batch and
heads for ``null_kv``.  The backward runs in exact fp32 whatever ``hip_precision`` the forward used (gradients of arbitrary
magnitude would need the power-of-two scaling the UNet's backward program carries, train.Backward.gscale)."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from .unet import LN_EPS, _Packed, _Pad, _ptr, default_precision, padded_head_dim

COLSUM_CHUNKS = 256          # row chunks of sgd_colsum's two-stage reduction


class _CrossAttnFn(torch.autograd.Function):
    """y = module(x, context, mask) with the module's parameters as autograd inputs (``loss.backward()`` fills their
    ``.grad`` like the reference's autograd does)"""

    @staticmethod
    def forward(ctx, mod, mask, x, context, *params):
        tape = {}
        y = mod._run(x.detach(), context.detach(), mask, tape)
        ctx.mod, ctx.tape = mod, tape
        ctx.save_for_backward(x, context, *params)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, context, *params = ctx.saved_tensors            # (raises if one of them was modified in place since)
        names = [n for n, _ in ctx.mod.named_parameters()]
        grads = ctx.mod._backward(ctx.tape, gy, dict(zip(names, params)))
        need = ctx.needs_input_grad
        out = [None, None, grads["x"] if need[2] else None, grads["context"] if need[3] else None]
        out += [grads[n] if need[4 + i] else None for i, n in enumerate(names)]
        return tuple(out)


class _LN(nn.Module):
    """attention_ldm.LayerNorm (:160-167): trainable gamma, beta is a zero BUFFER"""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))


class _Linear(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(nn.Linear(i, o, bias=False).weight.detach().clone())


class CrossAttention(nn.Module):
    LINEAR = False

    def __init__(self, dim, *, context_dim=None, dim_head=64, heads=8, norm_context=False):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head, self.dim = heads, dim_head, dim
        inner = dim_head * heads
        self.context_dim = dim if context_dim is None else context_dim
        self.norm = _LN(dim)
        self.norm_context = _LN(self.context_dim) if norm_context else nn.Identity()
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = _Linear(dim, inner)
        self.to_kv = _Linear(self.context_dim, inner * 2)
        self.to_out = nn.Sequential(_Linear(inner, dim), _LN(dim))
        self.hip_precision = default_precision()
        self._packs = {}

    # ---- packed operators (zero-padded head layout when dim_head has no attention-core instance, like unet._build_attn)
    def _pack(self, key, param, prec, pad):
        pk = self._packs.get((key, prec))
        if pk is None:
            pk = self._packs[(key, prec)] = _Packed([param], 1, prec, pad)
        pk.refresh(torch.cuda.current_stream().cuda_stream)
        return pk

    def _igemm(self, lib, x, cin, y, cout, pk, m, prec, ln=None, orows=(0, 0, 0)):
        a = L.IgemmArgs()
        a.x0, a.c0, a.mode, a.m, a.stride = x.data_ptr(), cin, L.MODE_FLAT, m, 1
        if ln is not None:
            stats, gamma, beta = ln
            a.pro, a.pa, a.pb, a.pc = L.PRO_LN_ROW, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr()
        a.w, a.cin_p, a.cout_p, a.w_scale_inv = pk.buf.data_ptr(), pk.cin_p, pk.cout_p, pk.scale_ptr
        a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cout, cout, prec
        a.orows_in, a.orows_out, a.orow_off = orows
        L.check(lib.sgd_igemm(C.byref(a), torch.cuda.current_stream().cuda_stream), "sgd_igemm")

    def forward(self, x, context, mask=None):
        if x.device.type != "cuda":
            raise RuntimeError("sgdm_amd attention_ldm modules run on the MI355X HIP path only; there is no CPU fallback")
        if torch.is_grad_enabled() and (x.requires_grad or context.requires_grad
                                        or any(p.requires_grad for p in self.parameters())):
            return _CrossAttnFn.apply(self, mask, x, context, *self.parameters())
        return self._run(x, context, mask, None)

    def _run(self, x, context, mask, tape):
        """the forward launch sequence; tape (dict or None): keep what the backward reads"""
        lib = L.load()
        st = torch.cuda.current_stream().cuda_stream
        prec = L.PREC_BY_NAME[self.hip_precision]
        b, n, dim = x.shape
        m = context.shape[1]
        heads, d = self.heads, self.dim_head
        dp = padded_head_dim(d)
        inner, J = heads * dp, m + 1
        x = x.contiguous().float()
        context = context.contiguous().float()
        dev = x.device
        qmap = [h * dp + i for h in range(heads) for i in range(d)]
        kvmap = qmap + [inner + j for j in qmap]                       # [k (all heads) | v (all heads)], padded per head
        padq = _Pad(rows=qmap, n_rows=inner) if dp != d else None
        padkv = _Pad(rows=kvmap, n_rows=2 * inner) if dp != d else None
        padout = _Pad(cols=qmap, n_cols=inner) if dp != d else None
        # q = to_q(LN(x))
        stx = torch.empty(b * n, 2, device=dev)
        L.check(lib.sgd_ln_stats(_ptr(x), b * n, dim, LN_EPS, _ptr(stx), st), "sgd_ln_stats")
        q = torch.empty(b, n, inner, device=dev)
        self._igemm(lib, x, dim, q, inner, self._pack("q", self.to_q.weight, prec, padq), b * n, prec,
                    ln=(stx, self.norm.gamma, self.norm.beta))
        # [null | to_kv(norm_context(context))] rows: k of head h at h*dp, v at inner + h*dp
        kv = torch.empty(b, J, 2 * inner, device=dev)
        ln_c = None
        if isinstance(self.norm_context, _LN):
            stc = torch.empty(b * m, 2, device=dev)
            L.check(lib.sgd_ln_stats(_ptr(context), b * m, self.context_dim, LN_EPS, _ptr(stc), st), "sgd_ln_stats")
            ln_c = (stc, self.norm_context.gamma, self.norm_context.beta)
        self._igemm(lib, context, self.context_dim, kv, 2 * inner, self._pack("kv", self.to_kv.weight, prec, padkv), b * m,
                    prec, ln=ln_c, orows=(m, J, 1))
        null = torch.zeros(2, heads, dp, device=dev)
        null[:, :, :d] = self.null_kv.detach().float()[:, None, :]
        kv[:, 0, :] = null.reshape(-1)                                   # repeat_many(null_kv, 'd -> b h 1 d') (:230)
        kmask = None
        if mask is not None:
            kmask = torch.ones(b, J, dtype=torch.uint8, device=dev)      # F.pad(mask, (1, 0), value=True) (:244)
            kmask[:, 1:] = mask.to(torch.uint8)
        # (linear core on padded heads: it writes the d real columns of each head only)
        att = (torch.zeros if (self.LINEAR and dp != d) else torch.empty)(b, n, inner, device=dev)
        kp, vp = _ptr(kv), C.c_void_p(kv.data_ptr() + 4 * inner)
        lse = torch.empty(b, heads, n, device=dev) if (tape is not None and not self.LINEAR) else None
        lp = _ptr(lse) if lse is not None else None
        if self.LINEAR:
            # the feature softmax runs over the TRUE head width d; heads are dp apart
            L.check(lib.sgd_linear_attention(_ptr(q), inner, dp, kp, vp, 2 * inner, dp,
                                             _ptr(kmask) if kmask is not None else None, b, heads, n, J, d, self.scale,
                                             _ptr(att), inner, st), "sgd_linear_attention")
        elif kmask is not None:
            L.check(lib.sgd_attention_masked(_ptr(q), inner, dp, kp, vp, 2 * inner, dp, _ptr(kmask), b, heads, n, J, dp,
                                             self.scale, _ptr(att), inner, lp, st), "sgd_attention_masked")
        else:
            fn = lib.sgd_attention_split if prec == L.PREC_F16X3 else lib.sgd_attention
            L.check(fn(_ptr(q), inner, dp, kp, vp, 2 * inner, dp, b, heads, n, J, dp, self.scale, _ptr(att), inner, lp, st),
                    "sgd_attention")
        o = torch.empty(b, n, dim, device=dev)
        self._igemm(lib, att, inner, o, dim, self._pack("out", self.to_out[0].weight, prec, padout), b * n, prec)
        y = torch.empty(b, n, dim, device=dev)
        L.check(lib.sgd_ln_apply(_ptr(o), _ptr(self.to_out[1].gamma), _ptr(self.to_out[1].beta), None, b * n, dim, LN_EPS,
                                 _ptr(y), st), "sgd_ln_apply")
        if tape is not None:
            tape.update(x=x, context=context, stx=stx, ln_c=ln_c, q=q, kv=kv, kmask=kmask, att=att, lse=lse, o=o,
                        pads=(padq, padkv, padout), qmap=qmap)
        return y

    # ---- backward (exact fp32; see the module docstring)
    def _dgrad(self, lib, st, g, w_fwd, param, cout_fwd, cin_fwd, rows, out=None):
        """out (+)= g [rows, cout_fwd] . w_fwd [cout_fwd, cin_fwd]: the forward kernel on the adjoint-packed weight"""
        from .train import _PackedAdj
        pk = _PackedAdj([param], lambda: w_fwd, cout_fwd, cin_fwd, 1, L.PREC_F32, g.device)
        pk.refresh(st)
        acc = out is not None
        y = out if acc else torch.empty(rows, cin_fwd, device=g.device)
        a = L.IgemmArgs()
        a.x0, a.c0, a.mode, a.m, a.stride = g.data_ptr(), cout_fwd, L.MODE_FLAT, rows, 1
        a.w, a.cin_p, a.cout_p, a.w_scale_inv = pk.buf.data_ptr(), pk.cin_p, pk.cout_p, pk.scale_ptr
        a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cin_fwd, cin_fwd, L.PREC_F32
        if acc:
            a.res, a.res_mode = y.data_ptr(), L.RS_NONE
        L.check(lib.sgd_igemm(C.byref(a), st), "sgd_igemm (input gradient)")
        return y

    def _wgrad(self, lib, st, xin, cin, ln, g, cout, rows):
        """dW [cout, cin] = g^T . act(xin) with act = the forward launch's LayerNorm-row prologue (or none)"""
        from .train import wgrad_ksplit
        a = L.IgemmArgs()
        a.x0, a.c0, a.mode, a.m, a.stride, a.prec = xin.data_ptr(), cin, L.MODE_FLAT, rows, 1, L.PREC_F32
        if ln is not None:
            stats, gamma, beta = ln
            a.pro, a.pa, a.pb, a.pc = L.PRO_LN_ROW, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr()
        ksplit = wgrad_ksplit(1, cout, cin, rows)
        slabs = torch.empty(ksplit, 1, cout, cin, device=g.device)
        L.check(lib.sgd_wgrad(C.byref(a), _ptr(g), cout, cout, _ptr(slabs), ksplit, None, st), "sgd_wgrad")
        dw = torch.empty(cout, cin, device=g.device)
        L.check(lib.sgd_wgrad_reduce(_ptr(slabs), ksplit, 1, cout, cin, _ptr(dw), 0, 1.0, st), "sgd_wgrad_reduce")
        return dw

    def _colsum(self, lib, st, g_ptr, rows, c, ld, dev):
        out = torch.empty(c, device=dev)
        work = torch.empty(COLSUM_CHUNKS, c, device=dev)
        L.check(lib.sgd_colsum(g_ptr, rows, c, ld, _ptr(out), 0, 1.0, _ptr(work), COLSUM_CHUNKS, st), "sgd_colsum")
        return out

    def _ln_bwd(self, lib, st, xin, g, gamma, rows, c):
        """(dL/dx, dL/dgamma) of y = LN(x) * gamma + beta (beta is a buffer: attention_ldm.py:160-167)"""
        dx, gxh = torch.empty(rows, c, device=g.device), torch.empty(rows, c, device=g.device)
        L.check(lib.sgd_ln_bwd(_ptr(xin), _ptr(g), _ptr(gamma), rows, c, LN_EPS, _ptr(dx), 0, _ptr(gxh), None, st), "sgd_ln_bwd")
        return dx, self._colsum(lib, st, _ptr(gxh), rows, c, c, g.device)

    def _backward(self, tape, gy, P):
        """adjoint of _run: {"x", "context", <parameter name>: gradient}.  P: parameter name -> the tensor the forward read."""
        lib = L.load()
        st = torch.cuda.current_stream().cuda_stream
        x, context, q, kv, att, kmask = tape["x"], tape["context"], tape["q"], tape["kv"], tape["att"], tape["kmask"]
        padq, padkv, padout = tape["pads"]
        b, n, dim = x.shape
        m, cdim = context.shape[1], self.context_dim
        heads, d = self.heads, self.dim_head
        dp = padded_head_dim(d)
        inner, J, rows = heads * dp, m + 1, b * n
        dev = x.device
        gy = gy.contiguous().float()
        grads = {}
        w_of = lambda name, pad: P[name].detach().float() if pad is None else pad.apply(P[name].detach().float())
        unpad = lambda dw, pad: dw if pad is None else pad.gather(dw).contiguous()
        # to_out: LayerNorm, then the projection
        go, grads["to_out.1.gamma"] = self._ln_bwd(lib, st, tape["o"], gy, P["to_out.1.gamma"], rows, dim)
        gatt = self._dgrad(lib, st, go, w_of("to_out.0.weight", padout), P["to_out.0.weight"], dim, inner, rows)
        grads["to_out.0.weight"] = unpad(self._wgrad(lib, st, att, inner, None, go, dim, rows), padout)
        # the core: gq with q's layout, gkv = [dk (all heads) | dv (all heads)] with kv's (zeros where the linear core on
        # padded heads writes nothing)
        alloc = torch.zeros if (self.LINEAR and dp != d) else torch.empty
        gq, gkv = alloc(b, n, inner, device=dev), alloc(b, J, 2 * inner, device=dev)
        kp, vp = _ptr(kv), C.c_void_p(kv.data_ptr() + 4 * inner)
        gkp, gvp = _ptr(gkv), C.c_void_p(gkv.data_ptr() + 4 * inner)
        mp = _ptr(kmask) if kmask is not None else None
        if self.LINEAR:
            L.check(lib.sgd_linear_attention_bwd(_ptr(q), inner, dp, kp, vp, 2 * inner, dp, mp, _ptr(gatt), inner, b, heads, n,
                                                 J, d, self.scale, _ptr(gq), gkp, gvp, st), "sgd_linear_attention_bwd")
        else:
            dvec = torch.empty(b, heads, n, device=dev)
            if kmask is not None:
                L.check(lib.sgd_attention_masked_bwd(_ptr(q), inner, dp, kp, vp, 2 * inner, dp, mp, _ptr(att), inner,
                                                     _ptr(gatt), inner, _ptr(tape["lse"]), _ptr(dvec), b, heads, n, J, dp,
                                                     self.scale, _ptr(gq), gkp, gvp, st), "sgd_attention_masked_bwd")
            else:
                L.check(lib.sgd_attention_bwd(_ptr(q), inner, dp, kp, vp, 2 * inner, dp, _ptr(att), inner, _ptr(gatt), inner,
                                              _ptr(tape["lse"]), _ptr(dvec), b, heads, n, J, dp, self.scale, _ptr(gq), gkp,
                                              gvp, st), "sgd_attention_bwd")
        # null key / value (row 0 of every batch element, the same [d] vector for every head, :230): sum over batch, then heads
        gnull = self._colsum(lib, st, _ptr(gkv), b, 2 * inner, J * 2 * inner, dev)             # [k | v] x heads x dp
        gn = torch.empty(2, dp, device=dev)
        work = torch.empty(COLSUM_CHUNKS, dp, device=dev)
        for i in range(2):
            L.check(lib.sgd_colsum(C.c_void_p(gnull.data_ptr() + 4 * i * inner), heads, dp, dp,
                                   C.c_void_p(gn.data_ptr() + 4 * i * dp), 0, 1.0, _ptr(work), COLSUM_CHUNKS, st), "sgd_colsum")
        grads["null_kv"] = gn[:, :d].contiguous()
        # to_q on LN(x)
        ln_x = (tape["stx"], P["norm.gamma"], self.norm.beta)
        gxn = self._dgrad(lib, st, gq, w_of("to_q.weight", padq), P["to_q.weight"], inner, dim, rows)
        grads["to_q.weight"] = unpad(self._wgrad(lib, st, x, dim, ln_x, gq, inner, rows), padq)
        grads["x"], grads["norm.gamma"] = self._ln_bwd(lib, st, x, gxn, P["norm.gamma"], rows, dim)
        grads["x"] = grads["x"].view(b, n, dim)
        # to_kv on norm_context(context): the gradient rows of the context keys / values (row 0 is the null pair)
        gkc = gkv[:, 1:, :].contiguous()
        ln_c = None
        if tape["ln_c"] is not None:
            ln_c = (tape["ln_c"][0], P["norm_context.gamma"], self.norm_context.beta)
        gcn = self._dgrad(lib, st, gkc, w_of("to_kv.weight", padkv), P["to_kv.weight"], 2 * inner, cdim, b * m)
        grads["to_kv.weight"] = unpad(self._wgrad(lib, st, context, cdim, ln_c, gkc, 2 * inner, b * m), padkv)
        if ln_c is not None:
            gcn, grads["norm_context.gamma"] = self._ln_bwd(lib, st, context, gcn, P["norm_context.gamma"], b * m, cdim)
        grads["context"] = gcn.view(b, m, cdim)
        return grads


class LinearCrossAttention(CrossAttention):
    """attention_ldm.py:261-298.  (The reference's masked branch broadcasts a [b, n, 1] mask against [(b h), n, d] keys, which
    only works for heads == 1; the HIP core applies the [b, keys] mask to every head -- identical where the reference runs.)"""
    LINEAR = True
