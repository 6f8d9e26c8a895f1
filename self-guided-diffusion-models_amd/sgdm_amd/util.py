"""Small host helpers of the boundary (reference diffusion_utils/util.py:70-100,254-268 and
diffusion_utils/lr_scheduler.py:36-98)."""
import importlib

import torch


def instantiate_from_config(config):
    assert "target" in config
    module, cls = config["target"].rsplit(".", 1)
    return getattr(importlib.import_module(module, package=None), cls)(**config.get("params", dict()))


class LambdaLinearScheduler:
    """lr_scheduler.py:36-98: linear warm-up to f_max, then linear f_max -> f_min over the cycle"""

    def __init__(self, warm_up_steps, f_min, f_max, f_start, cycle_lengths, verbosity_interval=0):
        assert len(warm_up_steps) == len(f_min) == len(f_max) == len(f_start) == len(cycle_lengths)
        self.lr_warm_up_steps, self.f_start, self.f_min, self.f_max = warm_up_steps, f_start, f_min, f_max
        self.cycle_lengths = cycle_lengths
        self.cum_cycles = [0]
        for c in cycle_lengths:
            self.cum_cycles.append(self.cum_cycles[-1] + c)
        self.last_f = 0.0

    def find_in_interval(self, n):
        for interval, cl in enumerate(self.cum_cycles[1:]):
            if n <= cl:
                return interval

    def schedule(self, n, **kwargs):
        cycle = self.find_in_interval(n)
        n = n - self.cum_cycles[cycle]
        if n < self.lr_warm_up_steps[cycle]:
            f = (self.f_max[cycle] - self.f_start[cycle]) / self.lr_warm_up_steps[cycle] * n + self.f_start[cycle]
        else:
            f = self.f_min[cycle] + (self.f_max[cycle] - self.f_min[cycle]) * (self.cycle_lengths[cycle] - n) / (
                self.cycle_lengths[cycle])
        self.last_f = f
        return f

    def __call__(self, n, **kwargs):
        return self.schedule(n, **kwargs)


def slerp_batch_torch(val, low, high):
    """diffusion_utils/util.py:48-60: spherical interpolation, val [K], low / high [1, C] -> [K, C]"""
    assert len(low.shape) == 2
    low_norm = low / torch.norm(low, dim=1, keepdim=True)
    high_norm = high / torch.norm(high, dim=1, keepdim=True)
    omega = torch.acos((low_norm * high_norm).sum(1))
    so = torch.sin(omega)
    return (torch.sin((1.0 - val) * omega) / so).unsqueeze(1) * low + (torch.sin(val * omega) / so).unsqueeze(1) * high


def batch_to_conditioninterp(cond_tensor, interp_num=9, samples=10, is_slerp=True):
    """eval/papervis_utils.py:362-394 (batch_to_conditioninterp_papervis): for i < samples interpolate cond[i] -> cond[i+1]
    in `interp_num` steps; result [(samples * interp_num), C] (host-side guidance preparation, a few rows)"""
    batch_size = len(cond_tensor)
    if batch_size < interp_num:
        interp_num = batch_size
    out = []
    for i in range(samples):
        c1, c2 = cond_tensor[i].reshape(1, -1), cond_tensor[i + 1].reshape(1, -1)
        lin_w = torch.linspace(0, 1, interp_num).to(cond_tensor.device)
        if is_slerp:
            feat = slerp_batch_torch(lin_w, c1, c2)
        else:
            feat = c1 * lin_w.reshape(-1, 1) + c2 * (1 - lin_w.reshape(-1, 1))
        out.append(feat.unsqueeze(0))
    return torch.cat(out, 0).reshape(-1, out[0].shape[-1])
