"""``dynamic.ema.LitEma`` drop-in (reference dynamic/ema.py:5-76): same buffers (dot-stripped names,
``decay``, ``num_updates``), same decay warm-up; the update itself is one multi-tensor pass per call
(``torch._foreach``) instead of one kernel per parameter.  Inside a training loop the shadow update normally does not
go through ``forward`` at all: ``optim.FusedAdamWEma`` folds it into the optimizer's single launch
(``sgd_adamw_ema_step``) and only advances ``num_updates`` here."""
import torch
from torch import nn


class LitEma(nn.Module):
    def __init__(self, model, decay=0.9999, use_num_upates=True):
        super().__init__()
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.m_name2s_name = {}
        self.register_buffer("decay", torch.tensor(decay, dtype=torch.float32))
        self.register_buffer("num_updates", torch.tensor(0, dtype=torch.int) if use_num_upates
                             else torch.tensor(-1, dtype=torch.int))
        for name, p in model.named_parameters():
            if p.requires_grad:
                s_name = name.replace(".", "")
                self.m_name2s_name[name] = s_name
                self.register_buffer(s_name, p.clone().detach().data)
        self.collected_params = []

    def forward(self, model):
        decay = self.decay
        if self.num_updates >= 0:
            self.num_updates += 1
            decay = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        one_minus_decay = float(1.0 - decay)
        with torch.no_grad():
            m_param = dict(model.named_parameters())
            shadow = dict(self.named_buffers())
            ps, ss = [], []
            for key, p in m_param.items():
                if p.requires_grad:
                    ps.append(p.detach())
                    ss.append(shadow[self.m_name2s_name[key]])
                else:
                    assert key not in self.m_name2s_name
            # shadow -= (1 - decay) * (shadow - p)
            diff = torch._foreach_sub(ss, ps)
            torch._foreach_add_(ss, diff, alpha=-one_minus_decay)

    def copy_to(self, model):
        m_param = dict(model.named_parameters())
        shadow = dict(self.named_buffers())
        with torch.no_grad():
            for key in m_param:
                if m_param[key].requires_grad:
                    m_param[key].copy_(shadow[self.m_name2s_name[key]])
                else:
                    assert key not in self.m_name2s_name

    def store(self, parameters):
        self.collected_params = [param.clone() for param in parameters]

    def restore(self, parameters):
        with torch.no_grad():
            for c_param, param in zip(self.collected_params, parameters):
                param.copy_(c_param)
