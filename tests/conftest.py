import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "self-guided-diffusion-models_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a GPU test on a box without a GPU is an error of selection, not a skip
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b):
    """max |a-b| / max |b|  (the "max-abs error / max-abs(ref)" column of SURVEY Appendix C)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elem_rel(a, b, floor=1e-2):
    """element-wise relative error with a floor: max_i |a_i - b_i| / max(|b_i|, floor * max|b|).  max_rel (above) is an absolute
    error in units of the tensor's largest element; this one also holds every element whose magnitude is at least `floor` of
    the maximum to a RELATIVE bound (VERDICT round 5, weak #1a).  An fp32 reorder moves any element by ~1e-6 of the tensor's
    scale, so for the smallest elements counted the bound reads 1 / floor times max_rel's."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    den = b.abs().clamp_min(floor * float(b.abs().max().clamp_min(1e-30)))
    return float(((a - b).abs() / den).max())


def cfg_from_index(entry):
    """unet_index.json entry (reference ctor kwargs) -> oracle cfg."""
    from oracle.unet_ref import make_cfg
    kw = entry["ctor"]
    return make_cfg(entry["kind"], kw["image_size"], kw["in_channels"], kw["out_channels"],
                    kw["model_channels"], kw["num_res_blocks"], kw["channel_mult"],
                    kw["attention_resolutions"], kw["num_heads"],
                    use_scale_shift_norm=kw["use_scale_shift_norm"],
                    resblock_updown=kw.get("resblock_updown", False), dropout=kw["dropout"],
                    cond_dim=kw["cond_dim"], condition_method=kw["condition_method"],
                    layout_dim=entry["layout_dim"], cond_token_num=kw.get("cond_token_num", 0),
                    context_dim=kw.get("context_dim"),
                    use_cls_token_as_pooled=kw.get("use_cls_token_as_pooled", True),
                    use_spatial_transformer=kw.get("use_spatial_transformer", False),
                    transformer_depth=kw.get("transformer_depth", 1),
                    use_new_attention_order=kw.get("use_new_attention_order", False))
