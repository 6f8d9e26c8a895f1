"""Oracle: functional CPU restatement of the reference UNet denoisers.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Plain PyTorch-CPU fp32,
NCHW, driven by a reference-format ``state_dict`` (name -> tensor) and a small
``cfg`` dict holding the reference constructor arguments.

Reference followed (all under /root/reference):
  unet_fast    dynamic/diffusionmodules/openaimodel.py     ctor :496-835, forward :904-956
  unetca_fast  dynamic/diffusionmodules/openaimodel_ca.py  ctor :479-836, forward :917-1033
  ResBlock     openaimodel.py:207-320 (== openaimodel_ca.py:184-297)
  Attention    openaimodel.py:323-424 (AttentionBlock + QKVAttentionLegacy)
  Attention_LR dynamic/crossattetion_lr.py:36-142
  helpers      dynamic/diffusionmodules/util.py:151-171 (timestep_embedding), :199-216 (GroupNorm32)
"""
import math

import torch
import torch.nn.functional as F

GN_GROUPS = 32     # util.py:205  normalization() == GroupNorm32(32, C)
GN_EPS = 1e-5      # nn.GroupNorm default
LN_EPS = 1e-5      # F.layer_norm / nn.LayerNorm default
NUM_TIME_TOKENS = 8   # openaimodel_ca.py:585
NUM_COND_TOKENS = 8   # openaimodel_ca.py:600


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
def make_cfg(kind, image_size, in_channels=3, out_channels=3, model_channels=128,
             num_res_blocks=2, channel_mult=(1, 2, 4), attention_resolutions=(4,),
             num_heads=8, num_head_channels=-1, use_scale_shift_norm=True,
             resblock_updown=None, conv_resample=True, dropout=0.0,
             cond_dim=0, condition_method=None, layout_dim=0, scale_type="imagen",
             cond_token_num=0, context_dim=None, use_cls_token_as_pooled=True,
             use_spatial_transformer=False, transformer_depth=1, use_new_attention_order=False):
    """Collect the reference ctor kwargs (config/dynamic/unet_fast.yaml:3-19,
    config/dynamic/unetca_fast.yaml:6-31).  ``layout_dim`` stands for
    ``condition.<condition_method>.layout_dim`` (config/condition/default.yaml)."""
    assert kind in ("unet_fast", "unetca_fast")
    if resblock_updown is None:
        # unet_fast.yaml:13 sets it; unetca_fast.yaml leaves the ctor default False
        resblock_updown = kind == "unet_fast"
    return dict(kind=kind, image_size=image_size, in_channels=in_channels,
                out_channels=out_channels, model_channels=model_channels,
                num_res_blocks=num_res_blocks, channel_mult=tuple(channel_mult),
                attention_resolutions=tuple(attention_resolutions), num_heads=num_heads,
                num_head_channels=num_head_channels,
                use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                conv_resample=conv_resample, dropout=dropout, cond_dim=cond_dim or 0,
                condition_method=condition_method, layout_dim=layout_dim,
                scale_type=scale_type, cond_token_num=cond_token_num,
                context_dim=context_dim, use_cls_token_as_pooled=use_cls_token_as_pooled,
                use_spatial_transformer=use_spatial_transformer, transformer_depth=transformer_depth,
                use_new_attention_order=bool(use_new_attention_order))


def _layout_channels(cfg):
    m = cfg["condition_method"]
    if cfg["kind"] == "unet_fast":
        return cfg["layout_dim"] if m == "clusterlayout" else 0          # openaimodel.py:623-630
    return cfg["layout_dim"] if m in ("clusterlayout", "stegoclusterlayout", "layout") else 0  # _ca.py:617-641


def _heads(cfg, ch):
    if cfg["num_head_channels"] == -1:
        return cfg["num_heads"]
    return ch // cfg["num_head_channels"]


def build_plan(cfg):
    """Block walk of the reference ctor (openaimodel.py:634-835, openaimodel_ca.py:645-836).

    Returns (input_blocks, middle_block, output_blocks); each block is a list
    of layer tuples:
      ("conv", cin, cout) | ("res", cin, cout, updown) | ("attn", ch, heads)
      | ("down", ch, use_conv) | ("up", ch, use_conv)
    """
    mc = cfg["model_channels"]
    cm = cfg["channel_mult"]
    nrb = cfg["num_res_blocks"]
    att = cfg["attention_resolutions"]
    in_ch = cfg["in_channels"] + _layout_channels(cfg)
    inp = [[("conv", in_ch, mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cm):
        for _ in range(nrb):
            layers = [("res", ch, mult * mc, None)]
            ch = mult * mc
            if ds in att:
                layers.append(("attn", ch, _heads(cfg, ch)))
            inp.append(layers)
            chans.append(ch)
        if level != len(cm) - 1:
            if cfg["resblock_updown"]:
                inp.append([("res", ch, ch, "down")])
            else:
                inp.append([("down", ch, cfg["conv_resample"])])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch, None), ("attn", ch, _heads(cfg, ch)), ("res", ch, ch, None)]
    out = []
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * mult, None)]
            ch = mc * mult
            if ds in att:
                layers.append(("attn", ch, _heads(cfg, ch)))
            if level and i == nrb:
                if cfg["resblock_updown"]:
                    layers.append(("res", ch, ch, "up"))
                else:
                    layers.append(("up", ch, cfg["conv_resample"]))
                ds //= 2
            out.append(layers)
    return inp, mid, out


# --------------------------------------------------------------------------
# parameter manifest (ordered like the reference module's state_dict())
# --------------------------------------------------------------------------
def _lin(p, i, o, bias=True):
    r = [(p + ".weight", (o, i), "param")]
    if bias:
        r.append((p + ".bias", (o,), "param"))
    return r


def _conv(p, i, o, k, dims=2):
    return [(p + ".weight", (o, i) + (k,) * dims, "param"), (p + ".bias", (o,), "param")]


def _norm(p, c):
    return [(p + ".weight", (c,), "param"), (p + ".bias", (c,), "param")]


def _layer_manifest(cfg, prefix, layer, emb_ch):
    kind = layer[0]
    ca = cfg["kind"] == "unetca_fast"
    if kind == "conv":
        return _conv(prefix, layer[1], layer[2], 3)
    if kind == "res":
        _, cin, cout, _ud = layer
        r = _norm(prefix + ".in_layers.0", cin) + _conv(prefix + ".in_layers.2", cin, cout, 3)
        r += _lin(prefix + ".emb_layers.1", emb_ch, 2 * cout if cfg["use_scale_shift_norm"] else cout)
        r += _norm(prefix + ".out_layers.0", cout) + _conv(prefix + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            r += _conv(prefix + ".skip_connection", cin, cout, 1)
        return r
    if kind == "attn":
        _, ch, heads = layer
        if not ca and cfg.get("use_spatial_transformer"):
            # SpatialTransformer(ch, num_heads, ch // num_heads, depth, context_dim): registration order of
            # dynamic/attention.py:246-258 (norm, proj_in, transformer_blocks, proj_out) and :198-215 (attn1, ff, attn2, norms)
            inner = heads * (ch // heads)
            r = _norm(prefix + ".norm", ch) + _conv(prefix + ".proj_in", ch, inner, 1)
            for i in range(cfg["transformer_depth"]):
                b = f"{prefix}.transformer_blocks.{i}"
                for att, kdim in ((".attn1", inner), (".attn2", cfg["context_dim"])):
                    blk = (_lin(b + att + ".to_q", inner, inner, bias=False) + _lin(b + att + ".to_k", kdim, inner, bias=False)
                           + _lin(b + att + ".to_v", kdim, inner, bias=False) + _lin(b + att + ".to_out.0", inner, inner))
                    if att == ".attn1":
                        blk += _lin(b + ".ff.net.0.proj", inner, 8 * inner) + _lin(b + ".ff.net.2", 4 * inner, inner)
                    r += blk
                r += _norm(b + ".norm1", inner) + _norm(b + ".norm2", inner) + _norm(b + ".norm3", inner)
            return r + _conv(prefix + ".proj_out", inner, ch, 1)
        if not ca:
            return (_norm(prefix + ".norm", ch) + _conv(prefix + ".qkv", ch, 3 * ch, 1, dims=1)
                    + _conv(prefix + ".proj_out", ch, ch, 1, dims=1))
        dh = ch // heads
        cd = cfg["context_dim"]
        return ([(prefix + ".null_kv", (2, dh), "param"),
                 (prefix + ".norm.gamma", (ch,), "param"), (prefix + ".norm.beta", (ch,), "buffer")]
                + _lin(prefix + ".to_q", ch, dh * heads, bias=False)
                + _lin(prefix + ".to_kv", ch, 2 * dh, bias=False)
                + _norm(prefix + ".to_context.0", cd) + _lin(prefix + ".to_context.1", cd, 2 * dh)
                + _lin(prefix + ".to_out.0", dh * heads, ch, bias=False)
                + [(prefix + ".to_out.1.gamma", (ch,), "param"), (prefix + ".to_out.1.beta", (ch,), "buffer")])
    if kind == "down":
        return _conv(prefix + ".op", layer[1], layer[1], 3) if layer[2] else []
    if kind == "up":
        return _conv(prefix + ".conv", layer[1], layer[1], 3) if layer[2] else []
    raise ValueError(kind)


def param_manifest(cfg):
    """Ordered [(name, shape, kind)] with kind in {param, frozen, buffer}; the
    order is that of ``reference_module.state_dict()`` (own parameters first,
    then children in registration order)."""
    mc = cfg["model_channels"]
    ted = 4 * mc
    cd = cfg["cond_dim"]
    S = cfg["image_size"]
    m = []
    if cfg["kind"] == "unet_fast":
        emb_ch = ted + (ted // 2 if cd > 0 else 0)
        if cd > 0:
            m.append(("null_cond_emb", (1, cd), "frozen"))                  # openaimodel.py:598-600
        if cfg["condition_method"] == "clusterlayout":
            m.append(("null_layout_emb", (1, 1, S, S), "frozen"))           # :624-626
        m += _lin("time_embed.0", mc, ted) + _lin("time_embed.2", ted, ted)  # :570-574
        if cd > 0:
            m += _lin("mlp_cond.0", cd, ted // 2) + _lin("mlp_cond.2", ted // 2, ted // 2)  # :603-607
    else:
        emb_ch = ted
        ctx = cfg["context_dim"]
        ctn = cfg["cond_token_num"]
        if ctn == 1:
            m.append(("null_cond_emb", (1, cd), "frozen"))                  # _ca.py:566-569
        elif ctn > 1:
            m.append(("null_cond_emb", (ctn, cd), "frozen"))
        if _layout_channels(cfg):
            m.append(("null_layout_emb", (1, 1, S, S), "frozen"))           # :617-641
        m += _lin("time_embed.0", mc, ted) + _lin("time_embed.2", ted, ted)
        m += _norm("norm_cond", ctx)                                         # :583
        m += _lin("to_time_tokens.0", mc, mc) + _lin("to_time_tokens.2", mc, ctx * NUM_TIME_TOKENS)
        if ctn > 0:
            m += _lin("cond_mlp.0", cd, ted) + _lin("cond_mlp.2", ted, ted)  # :594-598
            m += _lin("to_cond_tokens.0", cd, ctx * NUM_COND_TOKENS)         # :601-604
            mid = int(math.sqrt(ctx * cd))                                   # :605
            m += (_lin("to_cond_tokens_2d.0", cd, mid) + _lin("to_cond_tokens_2d.2", mid, mid)
                  + _lin("to_cond_tokens_2d.4", mid, mid) + _lin("to_cond_tokens_2d.6", mid, ctx))
    inp, mid_b, out = build_plan(cfg)
    for i, blk in enumerate(inp):
        for j, layer in enumerate(blk):
            m += _layer_manifest(cfg, f"input_blocks.{i}.{j}", layer, emb_ch)
    for j, layer in enumerate(mid_b):
        m += _layer_manifest(cfg, f"middle_block.{j}", layer, emb_ch)
    for i, blk in enumerate(out):
        for j, layer in enumerate(blk):
            m += _layer_manifest(cfg, f"output_blocks.{i}.{j}", layer, emb_ch)
    m += _norm("out.0", mc) + _conv("out.2", mc, cfg["out_channels"], 3)     # openaimodel.py:830-835
    return m


# --------------------------------------------------------------------------
# leaf ops
# --------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
    """util.py:151-171 (repeat_only=False)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(sd, p, x):
    return F.group_norm(x.float(), GN_GROUPS, sd[p + ".weight"], sd[p + ".bias"], GN_EPS)


def _linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _mlp2(sd, p, x, i0=0, i1=2):
    """Linear -> SiLU -> Linear (time_embed / mlp_cond / cond_mlp / to_time_tokens)."""
    return _linear(sd, f"{p}.{i1}", F.silu(_linear(sd, f"{p}.{i0}", x)))


def res_block(cfg, sd, p, x, emb, updown, dropout_mask=None):
    """openaimodel.py:300-320.  ``dropout_mask`` (optional, same shape as the
    second conv's input, already scaled by 1/(1-p)) injects the train-time
    dropout of out_layers[2] (openaimodel.py:272)."""
    h = F.silu(_gn(sd, p + ".in_layers.0", x))
    if updown == "down":                                   # :301-306 with Downsample(use_conv=False) :200
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    elif updown == "up":                                   # Upsample(use_conv=False) :151
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    emb_out = _linear(sd, p + ".emb_layers.1", F.silu(emb))[:, :, None, None]   # :262-268,309-311
    if cfg["use_scale_shift_norm"]:                        # :312-316
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = _gn(sd, p + ".out_layers.0", h) * (1 + scale) + shift
        h = F.silu(h)
    else:                                                  # :318-319
        h = F.silu(_gn(sd, p + ".out_layers.0", h + emb_out))
    if dropout_mask is not None:
        h = h * dropout_mask
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:              # :279-287 (1x1 conv) else Identity
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def attention_block(sd, p, x, heads, new_order=False):
    """AttentionBlock._forward + QKVAttentionLegacy (openaimodel.py:365-371, :403-420); new_order: QKVAttention
    (use_new_attention_order=True, openaimodel.py:350-352, :427-455) -- q | k | v split BEFORE the heads"""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(sd, p + ".norm", xf), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    length = qkv.shape[-1]
    ch = c // heads
    if new_order:
        q, k, v = (t.reshape(b * heads, ch, length) for t in qkv.chunk(3, dim=1))   # q|k|v, then heads (openaimodel.py:443-451)
    else:
        q, k, v = qkv.reshape(b * heads, ch * 3, length).split(ch, dim=1)   # legacy order: heads, then q|k|v
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, length)
    h = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(b, c, hh, ww)


def _cross_attention(sd, p, x, context, heads):
    """CrossAttention.forward (dynamic/attention.py:153-194), mask=None"""
    q = F.linear(x, sd[p + ".to_q.weight"])
    ctx = x if context is None else context
    k, v = F.linear(ctx, sd[p + ".to_k.weight"]), F.linear(ctx, sd[p + ".to_v.weight"])
    b, n, _ = q.shape
    d = q.shape[-1] // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)
    q, k, v = split(q), split(k), split(v)
    attn = (torch.einsum("bid,bjd->bij", q, k) * d ** -0.5).softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v).reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, heads * d)
    return F.linear(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def spatial_transformer(cfg, sd, p, x, heads, context=None):
    """SpatialTransformer.forward -> BasicTransformerBlock._forward -> GEGLU feed-forward
    (dynamic/attention.py:260-270, :217-221, :38-65); GroupNorm eps 1e-6 (:77-78)"""
    b, c, hh, ww = x.shape
    ln = lambda q, t: F.layer_norm(t, (t.shape[-1],), sd[q + ".weight"], sd[q + ".bias"], 1e-5)
    h = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    h = F.conv2d(h, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    h = h.reshape(b, h.shape[1], hh * ww).permute(0, 2, 1)
    for i in range(cfg["transformer_depth"]):
        k = f"{p}.transformer_blocks.{i}"
        h = _cross_attention(sd, k + ".attn1", ln(k + ".norm1", h), None, heads) + h
        h = _cross_attention(sd, k + ".attn2", ln(k + ".norm2", h), context, heads) + h
        a, gate = F.linear(ln(k + ".norm3", h), sd[k + ".ff.net.0.proj.weight"], sd[k + ".ff.net.0.proj.bias"]).chunk(2, dim=-1)
        h = F.linear(a * F.gelu(gate), sd[k + ".ff.net.2.weight"], sd[k + ".ff.net.2.bias"]) + h
    h = h.permute(0, 2, 1).reshape(b, -1, hh, ww)
    return F.conv2d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"]) + x


def attention_lr(sd, p, x, context, heads):
    """Attention_LR.forward (crossattetion_lr.py:81-142): multi-query attention
    over [context tokens | null kv | self tokens]."""
    b, c, w_, h_ = x.shape
    xs = x.permute(0, 2, 3, 1).reshape(b, w_ * h_, c)                       # 'b c w h -> b (w h) c'
    xn = F.layer_norm(xs, (c,), sd[p + ".norm.gamma"], sd[p + ".norm.beta"], LN_EPS)
    q = F.linear(xn, sd[p + ".to_q.weight"])
    kv = F.linear(xn, sd[p + ".to_kv.weight"])
    dh = kv.shape[-1] // 2
    k, v = kv[..., :dh], kv[..., dh:]
    q = q.reshape(b, w_ * h_, heads, dh).permute(0, 2, 1, 3) * (dh ** -0.5)   # :90-91
    nk = sd[p + ".null_kv"][0].expand(b, 1, dh)
    nv = sd[p + ".null_kv"][1].expand(b, 1, dh)
    k = torch.cat((nk, k), dim=-2)                                          # :95-97
    v = torch.cat((nv, v), dim=-2)
    if context is not None:                                                 # :101-105
        cn = F.layer_norm(context, (context.shape[-1],), sd[p + ".to_context.0.weight"],
                          sd[p + ".to_context.0.bias"], LN_EPS)
        ckv = F.linear(cn, sd[p + ".to_context.1.weight"], sd[p + ".to_context.1.bias"])
        k = torch.cat((ckv[..., :dh], k), dim=-2)
        v = torch.cat((ckv[..., dh:], v), dim=-2)
    sim = torch.einsum("bhid,bjd->bhij", q, k)                              # :115
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bjd->bhid", attn, v)                           # :137
    out = out.permute(0, 2, 1, 3).reshape(b, w_ * h_, heads * dh)
    out = F.linear(out, sd[p + ".to_out.0.weight"])
    out = F.layer_norm(out, (c,), sd[p + ".to_out.1.gamma"], sd[p + ".to_out.1.beta"], LN_EPS)
    return (xs + out).reshape(b, w_, h_, c).permute(0, 3, 1, 2)


def _run_block(cfg, sd, prefix, blk, h, emb, context, dropout_masks):
    ca = cfg["kind"] == "unetca_fast"
    for j, layer in enumerate(blk):
        p = f"{prefix}.{j}"
        kind = layer[0]
        if kind == "conv":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif kind == "res":
            dm = None if dropout_masks is None else dropout_masks.get(p)
            h = res_block(cfg, sd, p, h, emb, layer[3], dm)
        elif kind == "attn":
            if ca:
                h = attention_lr(sd, p, h, context, layer[2])
            elif cfg.get("use_spatial_transformer"):
                h = spatial_transformer(cfg, sd, p, h, layer[2])              # openaimodel.py:915: context is always None
            else:
                h = attention_block(sd, p, h, layer[2], cfg.get("use_new_attention_order", False))
        elif kind == "down":
            if layer[2]:                                    # conv stride 2 (openaimodel_ca.py:167-174)
                h = F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
            else:
                h = F.avg_pool2d(h, 2, 2)
        elif kind == "up":                                  # openaimodel_ca.py:128-131
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            if layer[2]:
                h = F.conv2d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        else:
            raise ValueError(kind)
    return h


# --------------------------------------------------------------------------
# whole-UNet forward
# --------------------------------------------------------------------------
def unet_forward(cfg, sd, x, t, cond=None, layout=None, drop_mask=None, dropout_masks=None):
    """One UNet evaluation -> eps_hat [B, out_channels, H, W].

    ``drop_mask`` is the boolean [B] classifier-free "drop the condition" mask
    (the reference draws it as ``uniform(B) < cond_drop_prob``, openaimodel.py:926-928;
    injecting it makes the evaluation deterministic: p=0 -> all False, p=1 -> all True).
    Follows openaimodel.py:904-956 / openaimodel_ca.py:917-1033.
    """
    B = x.shape[0]
    mc = cfg["model_channels"]
    if drop_mask is None:
        drop_mask = torch.zeros(B, dtype=torch.bool)
    t_emb = timestep_embedding(t, mc)
    emb = _mlp2(sd, "time_embed", t_emb)
    context = None
    if cfg["kind"] == "unet_fast":
        if cfg["cond_dim"] > 0:
            cond = cond.to(sd["null_cond_emb"].dtype)                       # :911
            cm = torch.where(drop_mask[:, None], sd["null_cond_emb"], cond)  # :929-931
            if cfg["condition_method"] == "clusterlayout":                  # :933-939
                lm = torch.where(drop_mask[:, None, None, None], sd["null_layout_emb"], layout)
                x = torch.cat((x, lm), dim=1)
            emb = torch.cat((emb, _mlp2(sd, "mlp_cond", cm)), dim=-1)       # :941-942
    else:
        time_tokens = _mlp2(sd, "to_time_tokens", t_emb).reshape(B, NUM_TIME_TOKENS, -1)   # :942
        ctn = cfg["cond_token_num"]
        if ctn == 0:                                                        # :944-958
            context = time_tokens
            if cfg["condition_method"] == "clusterlayout":
                raise NotImplementedError
            if cfg["condition_method"] == "layout":
                lm = torch.where(drop_mask[:, None, None, None], sd["null_layout_emb"], layout)
                x = torch.cat((x, lm), dim=1)
        elif ctn == 1:                                                      # :960-986
            assert cond.dim() == 2
            cm = torch.where(drop_mask[:, None], sd["null_cond_emb"], cond)
            cond_tokens = _linear(sd, "to_cond_tokens.0", cm).reshape(B, NUM_COND_TOKENS, -1)
            context = torch.cat([time_tokens, cond_tokens], 1)
            emb = emb + _mlp2(sd, "cond_mlp", cm)
            if cfg["condition_method"] in ("clusterlayout", "stegoclusterlayout"):
                lm = torch.where(drop_mask[:, None, None, None], sd["null_layout_emb"], layout)
                x = torch.cat((x, lm), dim=1)
        else:                                                               # :988-1012
            assert cond.dim() == 3
            cm = torch.where(drop_mask[:, None, None], sd["null_cond_emb"], cond)
            z = cm
            for i in (0, 2, 4):
                z = F.silu(_linear(sd, f"to_cond_tokens_2d.{i}", z))
            cond_tokens = _linear(sd, "to_cond_tokens_2d.6", z)
            context = torch.cat([time_tokens, cond_tokens], 1)
            pooled = cm[:, 0, :] if cfg["use_cls_token_as_pooled"] else cm.mean(dim=1)
            if cfg["condition_method"] == "clusterlayout":
                raise NotImplementedError
            emb = emb + _mlp2(sd, "cond_mlp", pooled)
        context = F.layer_norm(context, (context.shape[-1],), sd["norm_cond.weight"],
                               sd["norm_cond.bias"], LN_EPS)                # :1017

    inp, mid, out = build_plan(cfg)
    hs = []
    h = x.float()
    for i, blk in enumerate(inp):
        h = _run_block(cfg, sd, f"input_blocks.{i}", blk, h, emb, context, dropout_masks)
        hs.append(h)
    h = _run_block(cfg, sd, "middle_block", mid, h, emb, context, dropout_masks)
    for i, blk in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)                                 # :950 (h first, skip second)
        h = _run_block(cfg, sd, f"output_blocks.{i}", blk, h, emb, context, dropout_masks)
    h = F.silu(_gn(sd, "out.0", h))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)     # :830-835


def guided_score(cfg, eps_uncond, eps_cond, w):
    """get_guided_score (openaimodel.py:853-859)."""
    if cfg["scale_type"] == "imagen":
        return (1 - w) * eps_uncond + w * eps_cond
    if cfg["scale_type"] == "cfg":
        return (1 + w) * eps_cond - w * eps_uncond
    raise ValueError(cfg["scale_type"])


def forward_with_cond_scale(cfg, sd, x, t, cond_scale, cond=None, layout=None):
    """openaimodel.py:861-902 / openaimodel_ca.py:879-915.  The two fast paths
    take a python number (the _ca variant only an ``int``)."""
    B = x.shape[0]
    is_num = isinstance(cond_scale, (int, float)) and not isinstance(cond_scale, bool)
    if cfg["kind"] == "unetca_fast":
        is_num = isinstance(cond_scale, int) and not isinstance(cond_scale, bool)
    if is_num and cond_scale == 1:
        return unet_forward(cfg, sd, x, t, cond, layout, torch.zeros(B, dtype=torch.bool))
    if is_num and cond_scale == 0:
        return unet_forward(cfg, sd, x, t, cond, layout, torch.ones(B, dtype=torch.bool))
    dbl = lambda a: None if a is None else torch.cat((a, a), 0)
    mask = torch.cat((torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)))
    eps = unet_forward(cfg, sd, dbl(x), dbl(t), dbl(cond), dbl(layout), mask)
    eps_c, eps_u = torch.chunk(eps, 2, dim=0)
    return guided_score(cfg, eps_u, eps_c, cond_scale)
