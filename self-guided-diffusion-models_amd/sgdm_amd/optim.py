"""Fused optimizer step of the training loop: AdamW + LitEma in one HIP launch.

Replaces ``torch.optim.AdamW(params, lr, weight_decay)`` of ``configure_optimizers``
(reference lightning_module_common.py:20-42) and the per-step ``LitEma.forward`` (dynamic/ema.py:25-44), which
together issue ~1000 small kernels per step.  ``FusedAdamWEma`` is a ``torch.optim.Optimizer`` (param_groups,
state / state_dict in torch.optim.AdamW's format, LR schedulers such as the reference's LambdaLR work unchanged);
pass the LitEma instance to fold the shadow update into the same kernel and stop calling ``ema(model)``.
"""
import ctypes as C

import torch

from . import _lib as L

CHUNK = 4096


def chunk_table(numels):
    """prefix sum of ceil(n / CHUNK): block b of the launch serves the tensor t with start[t] <= b < start[t + 1]"""
    start = [0]
    for n in numels:
        start.append(start[-1] + (int(n) + CHUNK - 1) // CHUNK)
    return start


class FusedAdamWEma(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, ema=None, ema_model=None):
        """ema: a LitEma whose shadows follow ``ema_model``'s parameters (names resolve the shadow buffers)"""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.ema = ema
        self._shadow_of = {}
        if ema is not None:
            if ema_model is None:
                raise ValueError("ema_model is required with ema")
            shadow = dict(ema.named_buffers())
            for name, p in ema_model.named_parameters():
                if p.requires_grad:
                    self._shadow_of[id(p)] = shadow[ema.m_name2s_name[name]]
        self._chunks = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        omd = -1.0
        if self.ema is not None:
            ema = self.ema
            decay = float(ema.decay)
            if int(ema.num_updates) >= 0:
                ema.num_updates += 1
                n = int(ema.num_updates)
                decay = min(decay, (1 + n) / (10 + n))
            omd = 1.0 - decay
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"]]
            if not ps:
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise RuntimeError("FusedAdamWEma runs on the GPU only (there is no CPU fallback)")
            rows, step_t, keep, touched = [], None, [], []
            for p in ps:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdamWEma expects contiguous fp32 parameters")
                st = self.state[p]
                g = p.grad
                if g is not None:
                    if not st:
                        st["step"] = torch.tensor(0.0)
                        st["exp_avg"] = torch.zeros_like(p)
                        st["exp_avg_sq"] = torch.zeros_like(p)
                    st["step"] += 1
                    if step_t is not None and float(st["step"]) != step_t:
                        raise RuntimeError("FusedAdamWEma: parameters of one group must share their step count")
                    step_t = float(st["step"])
                    g = g if g.is_contiguous() else g.contiguous()
                    keep.append(g)                                # alive until the launch has consumed it; NOT optimizer
                                                                  # state (state_dict stays torch.optim.AdamW's format)
                    touched.append(p)
                sh = self._shadow_of.get(id(p))
                rows.append((p.data_ptr(), g.data_ptr() if g is not None else 0,
                             st["exp_avg"].data_ptr() if g is not None else 0,
                             st["exp_avg_sq"].data_ptr() if g is not None else 0,
                             sh.data_ptr() if sh is not None else 0, p.numel()))
            if step_t is None and omd < 0:
                continue
            key = (gi, tuple(r[5] for r in rows))
            if key not in self._chunks:
                start = chunk_table([r[5] for r in rows])
                self._chunks[key] = (torch.tensor(start, dtype=torch.int32, device=dev), start[-1])
            cstart, total = self._chunks[key]
            table = torch.tensor(rows, dtype=torch.int64).to(dev, non_blocking=True)
            b1, b2 = group["betas"]
            t = step_t if step_t is not None else 1.0
            # gate (ABI 18): the device's gradient-health flag -- the launch writes nothing while it is up, so gradients a
            # balanced-tail time-out poisoned never reach parameters, moments or EMA shadows (train.Backward.run raises it,
            # the engine's next poll_health() raises on the host and clears it)
            from .unet import grad_health
            L.check(lib.sgd_adamw_ema_step(C.c_void_p(table.data_ptr()), C.c_void_p(cstart.data_ptr()), len(rows), total,
                                           float(group["lr"]), 1.0 - b1, b2, 1.0 - b2, group["eps"], group["weight_decay"],
                                           1.0 - b1 ** t, 1.0 - b2 ** t, omd, C.c_void_p(grad_health(dev).data_ptr()),
                                           torch.cuda.current_stream().cuda_stream), "sgd_adamw_ema_step")
            self._table_keep = (table, keep)
            # The kernel writes the parameters through raw pointers, which autograd's version counters do not see.
            # Consumers that cache derived copies keyed on (data_ptr, _version) -- the UNet's igemm-packed forward and
            # adjoint weights (unet._Packed, train._PackedAdj) and the FiLM bias concat -- must observe the update.
            torch._C._increment_version(touched)
        return loss
