"""Plugin surface ``sg.params.condition_method`` -> denoiser kwargs.

Mirrors reference ``dynamic_input/condition.py:5-157`` (prepare_condition_kwargs,
prepare_denoise_fn_kwargs_4sharestep, randomsample_cond, prepare_denoise_fn_kwargs_4sampling):
same method names, same batch keys, same train/eval ``cond_drop_prob`` rule, same errors.
Table-driven instead of an if-ladder; pinned by tests/golden/condition_plugin.json.
"""

# condition_method -> (cond key or None, {how: layout key} or layout key or None)
_DIRECT = ("label", "attr", "feat", "knn_feat", "patchfeat", "centroid", "labelcentroid", "cluster", "clustermix",
           "clusterrandom", "labelcluster", "patchcluster")
_LAYOUT_BY_HOW = {"lost": "lostbboxmask", "oracle": "segmask", "stego": "stegomask"}
_RANDOMISABLE = {"label": "label_random", "cluster": "cluster_random", "centroid": "centroid_random",
                 "knn_feat": "knn_feat_random"}
_NOT_RANDOMISABLE = ("feat", "attr", "labelcluster", "labelcentroid", "clusterlayout", "stegoclusterlayout",
                     "clustermix", "clusterrandom", "layout", "patchcluster", "patchfeat")


def _f(t, pl_module):
    return t.float().to(pl_module.device)


def prepare_condition_kwargs(pl_module, batch_data):
    hp = pl_module.hparams
    method = hp.condition_method
    if method is not None:
        assert hp.cond_drop_prob > 0
        out = dict(cond_drop_prob=hp.cond_drop_prob if pl_module.training else 1.0)     # condition.py:11-13
    else:
        out = dict(cond_drop_prob=1.0)
    if method is None:
        out.update(cond=None)
    elif method in _DIRECT:
        out.update(cond=batch_data[method])
    elif method == "cluster_lookup":
        out.update(cond=None, image_batch_ids=batch_data["id"])
    elif method == "clusterlayout":
        key = _LAYOUT_BY_HOW.get(hp.condition.clusterlayout.how)
        if key is None:
            raise RuntimeError("unknown clusterlayout.how")
        out.update(cond=_f(batch_data["cluster"], pl_module), layout=_f(batch_data[key], pl_module))
    elif method == "layout":
        key = _LAYOUT_BY_HOW.get(hp.condition.layout.how)
        if key is None:
            raise RuntimeError("unknown layout.how")
        out.update(layout=_f(batch_data[key], pl_module))
    elif method == "stegoclusterlayout":
        out.update(cond=_f(batch_data["stego_attr"], pl_module), layout=_f(batch_data["stegomask"], pl_module))
    else:
        raise ValueError(method)
    return out


def prepare_denoise_fn_kwargs_4sharestep(pl_module, batch_data):
    return prepare_condition_kwargs(pl_module=pl_module, batch_data=batch_data)


def randomsample_cond(pl_module, data_dict, random_sample_condition):
    method = pl_module.hparams.condition_method
    if method is None or method in _NOT_RANDOMISABLE:
        if random_sample_condition:
            raise RuntimeError(f"random_sample_condition is not defined for condition_method={method}")
    elif method in _RANDOMISABLE:
        if random_sample_condition:
            data_dict[method] = data_dict[_RANDOMISABLE[method]]
    else:
        raise ValueError(method)
    return data_dict


def prepare_denoise_fn_kwargs_4sampling(pl_module, batch_data, sampling_kwargs, cond_scale):
    batch_data = randomsample_cond(pl_module, data_dict=batch_data,
                                   random_sample_condition=sampling_kwargs["random_sample_condition"])
    kw = prepare_denoise_fn_kwargs_4sharestep(pl_module, batch_data)
    kw.update(dict(cond_scale=cond_scale))
    kw.pop("cond_drop_prob")            # sampling never passes it (condition.py:154-155)
    return kw
