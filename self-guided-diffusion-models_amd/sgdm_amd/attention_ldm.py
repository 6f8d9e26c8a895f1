"""``dynamic.attention_ldm.CrossAttention`` / ``LinearCrossAttention`` (reference dynamic/attention_ldm.py:198-298,
SURVEY row A23) on the HIP kernels of the hot path.

Imagen-style per-head cross-attention with a learned null key/value in front of the context and an optional context
mask; the linear variant replaces softmax(q k^T) v by softmax_d(q) (softmax_keys(k)^T v).  Same constructor keywords,
parameter / buffer names and shapes (``norm.gamma``, ``norm.beta`` [buffer], ``norm_context.*``, ``null_kv``,
``to_q.weight``, ``to_kv.weight``, ``to_out.0.weight``, ``to_out.1.gamma`` / ``.beta``) and the same forward contract as
the reference classes, so state dicts interchange.  No shipped config instantiates them (only ``log`` is imported from
that module), so this is an inference path: LayerNorm statistics + LN-prologue projections (``sgd_ln_stats`` /
``sgd_igemm``), the MFMA attention core (``sgd_attention_split`` / ``sgd_attention`` / ``sgd_attention_masked``) or the
linear core (``sgd_linear_attention``), output projection and ``sgd_ln_apply``.  There is no CPU fallback and no
autograd through the module (it raises under grad mode with trainable parameters instead of silently detaching)."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from .unet import LN_EPS, _Packed, _Pad, _ptr, default_precision, padded_head_dim


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
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("attention_ldm.CrossAttention on the HIP path is inference-only (no shipped config "
                                      "instantiates it); call it under torch.no_grad()")
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
        if self.LINEAR:
            # the feature softmax runs over the TRUE head width d; heads are dp apart
            L.check(lib.sgd_linear_attention(_ptr(q), inner, dp, kp, vp, 2 * inner, dp,
                                             _ptr(kmask) if kmask is not None else None, b, heads, n, J, d, self.scale,
                                             _ptr(att), inner, st), "sgd_linear_attention")
        elif kmask is not None:
            L.check(lib.sgd_attention_masked(_ptr(q), inner, dp, kp, vp, 2 * inner, dp, _ptr(kmask), b, heads, n, J, dp,
                                             self.scale, _ptr(att), inner, None, st), "sgd_attention_masked")
        else:
            fn = lib.sgd_attention_split if prec == L.PREC_F16X3 else lib.sgd_attention
            L.check(fn(_ptr(q), inner, dp, kp, vp, 2 * inner, dp, b, heads, n, J, dp, self.scale, _ptr(att), inner, None, st),
                    "sgd_attention")
        o = torch.empty(b, n, dim, device=dev)
        self._igemm(lib, att, inner, o, dim, self._pack("out", self.to_out[0].weight, prec, padout), b * n, prec)
        y = torch.empty(b, n, dim, device=dev)
        L.check(lib.sgd_ln_apply(_ptr(o), _ptr(self.to_out[1].gamma), _ptr(self.to_out[1].beta), None, b * n, dim, LN_EPS,
                                 _ptr(y), st), "sgd_ln_apply")
        return y


class LinearCrossAttention(CrossAttention):
    """attention_ldm.py:261-298.  (The reference's masked branch broadcasts a [b, n, 1] mask against [(b h), n, d] keys, which
    only works for heads == 1; the HIP core applies the [b, keys] mask to every head -- identical where the reference runs.)"""
    LINEAR = True
