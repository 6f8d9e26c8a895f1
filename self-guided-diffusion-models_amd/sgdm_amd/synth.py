"""Deterministic synthetic weights and inputs (no datasets / checkpoints exist offline).

Used by the parity tests, ``tests/golden/make_golden.py`` and ``bench.py`` so that
every party (reference run, oracle, HIP path) sees bit-identical tensors without
committing multi-MB weight files: tensors are regenerated from ``(name, shape, seed)``
with numpy's frozen legacy ``RandomState`` stream.

Weight recipe (SURVEY.md 8(d) "Synthetic inputs", adapted): conv/linear weights
U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (PyTorch's default bound); the reference's
zero-initialised tensors (openaimodel.py:273-276,357,833-834) are drawn the same way,
otherwise a fresh UNet outputs exactly 0; norm gains 1+0.1 N(0,1); biases 0.05 N(0,1);
``null_kv`` N(0,1); the frozen ``null_*_emb`` tensors and the LayerNorm ``beta``
buffers stay zero as in the reference.
"""
import zlib

import numpy as np
import torch


_COND_INPUT_LAYERS = ("mlp_cond.0.weight", "cond_mlp.0.weight", "to_cond_tokens.0.weight",
                      "to_cond_tokens_2d.0.weight")


def tensor_from_seed(name, shape, seed=23):
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF)
    leaf = name.rsplit(".", 1)[-1]
    shape = tuple(shape)
    if leaf in ("null_cond_emb", "null_layout_emb", "beta"):
        a = np.zeros(shape)
    elif leaf == "null_kv":
        a = rs.standard_normal(shape)
    elif leaf == "gamma" or (leaf == "weight" and len(shape) == 1):
        a = 1.0 + 0.1 * rs.standard_normal(shape)
    elif leaf == "bias":
        a = 0.05 * rs.standard_normal(shape)
    elif leaf == "weight":
        bound = 1.0 / np.sqrt(float(np.prod(shape[1:])))
        if name in _COND_INPUT_LAYERS:
            bound = 0.5      # inputs are one-/n-hot rows: keep the guidance signal visible in eps
        a = rs.uniform(-bound, bound, size=shape)
    else:
        raise ValueError(f"no synthetic recipe for {name}")
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def weights_from_seed(manifest, seed=23):
    """manifest: iterable of (name, shape[, kind]) -> dict name -> float32 CPU tensor."""
    return {m[0]: tensor_from_seed(m[0], m[1], seed) for m in manifest}


def _gen(seed, tag):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(tag.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def synth_batch(condition_method, batch, image_size, cond_dim, layout_dim=0, channels=3, seed=23):
    """Synthetic guidance tensors with the dataset contract of SURVEY.md 8(a) A2.

    label/cluster       cond int64 one-hot [B,K]            (supervised_label.py:31-40, unsupervised_cluster.py:33-46)
    clusterlayout       cond f32 one-hot [B,K], layout f32 {0,1} box mask [B,1,S,S]   (complex_ds_common_util.py:151-162)
    stegoclusterlayout  layout f32 one-hot over L channels from an 8x8 label map upsampled
                        nearest, cond f32 n-hot of the labels present [B,L]            (complex_ds_common_util.py:118-133)
    """
    S = image_size
    out = {"image": torch.rand(batch, channels, S, S, generator=_gen(seed, "image")) * 2 - 1}
    if condition_method in ("label", "cluster"):
        idx = torch.randint(0, cond_dim, (batch,), generator=_gen(seed, "cond"))
        out["cond"] = torch.nn.functional.one_hot(idx, cond_dim).to(torch.int64)
    elif condition_method == "clusterlayout":
        idx = torch.randint(0, cond_dim, (batch,), generator=_gen(seed, "cond"))
        out["cond"] = torch.nn.functional.one_hot(idx, cond_dim).float()
        g = _gen(seed, "layout")
        lay = torch.zeros(batch, layout_dim, S, S)
        for b in range(batch):
            y0, x0 = (int(v) for v in torch.randint(0, S // 2, (2,), generator=g))
            hh, ww = (int(v) for v in torch.randint(S // 4, S // 2 + 1, (2,), generator=g))
            lay[b, :, y0:y0 + hh, x0:x0 + ww] = 1.0
        out["layout"] = lay
    elif condition_method in ("stegoclusterlayout", "layout"):
        g = _gen(seed, "layout")
        assert S % 8 == 0
        lab = torch.randint(0, layout_dim, (batch, 8, 8), generator=g)
        rep = S // 8
        lab = lab.repeat_interleave(rep, dim=1).repeat_interleave(rep, dim=2)
        out["layout"] = torch.nn.functional.one_hot(lab, layout_dim).permute(0, 3, 1, 2).float().contiguous()
        if condition_method == "stegoclusterlayout":
            out["cond"] = (out["layout"].sum(dim=(2, 3)) > 0).float()
    elif condition_method is None:
        pass
    else:
        raise ValueError(condition_method)
    return out
