"""Guidance-tensor ingest on the device (sgdm_amd/guidance.py): the compact forms (class / cluster
ids, uint8 label maps, box corners) expanded inside the boundary kernels must give bit-identical tensors / UNet outputs
to the reference's CPU-side expansion (dataset/transforms/complex_ds_common_util.py:103-162, dataset/ds_utils/
unsupervised_cluster.py:33-46), restated here with the reference's own torch formulas.  GPU only."""
import pytest
import torch
import torch.nn.functional as F

from conftest import max_rel
from test_hip_unet import build_model

pytestmark = pytest.mark.gpu


def _ref_onehot_mask(label_map, k):
    """stego_to_onehotmask (:118-123) per image: 255 -> 0, one_hot, 'w h c -> c w h'"""
    out = []
    for m in label_map:
        m = m.clone().long()
        m[m == 255] = 0
        out.append(F.one_hot(m, num_classes=k).permute(2, 0, 1))
    return torch.stack(out).float()


def _ref_nhot(label_map, k):
    """stegomask_to_attr_nhot (:126-133) per image"""
    return torch.stack([F.one_hot(torch.unique(m).long(), num_classes=k).sum(0) for m in label_map]).float()


def test_labelmap_expansions_bit_exact():
    from sgdm_amd import guidance as G
    g = torch.Generator().manual_seed(3)
    lm = torch.randint(0, 27, (5, 8, 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2).to(torch.uint8)
    lm[0, :3, :5] = 255                                         # the "background" code of the png masks
    lm[1] = 7                                                   # a single label
    ref_mask = _ref_onehot_mask(lm, 27)
    lm_nhot = lm.clone()
    lm_nhot[lm_nhot == 255] = 0                                 # the reference calls the n-hot on 255-free masks
    assert torch.equal(G.onehot_layout(lm.cuda(), 27).cpu(), ref_mask)
    assert torch.equal(G.stego_attr(lm_nhot.cuda(), 27).cpu(), _ref_nhot(lm_nhot, 27))


def test_box_expansion_bit_exact():
    from sgdm_amd import guidance as G
    boxes = torch.tensor([[3, 5, 40, 60], [0, 0, 64, 64], [10, 10, 10, 30], [63, 0, 64, 1]])
    ref = torch.zeros(4, 1, 64, 64)
    for i, b in enumerate(boxes.tolist()):
        ref[i, 0, b[1]:b[3], b[0]:b[2]] = 1                     # bboxmask[bbox[1]:bbox[3], bbox[0]:bbox[2]] = 1 (:157)
    assert torch.equal(G.box_layout(boxes.cuda(), 64, 64).cpu(), ref)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_unet_fast_cluster_ids_equal_onehot(prec):
    """cluster k=5000: ids in (8 bytes/sample) == one-hot rows in (40 KB/sample), incl. dropped rows; the gathered
    mlp_cond.0 is bit-identical to the dense product"""
    m, entry = build_model("uf_cluster5000_c32_s16", prec)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 5000, (2,), generator=g)
    x, t = torch.randn(2, 3, 16, 16, generator=g).cuda(), torch.tensor([999, 3]).cuda()
    onehot = F.one_hot(ids, 5000)
    for mask in ([False, False], [False, True], [True, True]):
        mk = torch.tensor(mask).cuda()
        with torch.no_grad():
            a = m(x, t, cond=onehot.cuda(), cond_drop_prob=0.0, cond_drop_mask=mk)[0]
            b = m(x, t, cond=ids.cuda(), cond_drop_prob=0.0, cond_drop_mask=mk)[0]
        assert torch.equal(a, b), mask
    with torch.no_grad():
        a = m.forward_with_cond_scale(x, t, cond_scale=2.0, cond=onehot.cuda())
        b = m.forward_with_cond_scale(x, t, cond_scale=2.0, cond=ids.cuda())
    assert torch.equal(a, b)
    # int64 one-hot ROWS (the dataset's format) run over their non-zero entries (sgd_linear_sparse_rows); the same rows as
    # float take the dense skinny GEMM: bit-identical for one-hot rows
    with torch.no_grad():
        c = m.forward_with_cond_scale(x, t, cond_scale=2.0, cond=onehot.float().cuda())
    assert torch.equal(a, c)


def test_mlp_cond_sparse_rows_kernel():
    """sgd_linear_sparse_rows against the dense product: one-hot rows bit-exact, multi-hot / counted / negative rows to fp32
    rounding, dropped rows take the null projection, an all-zero row gives the bias"""
    from sgdm_amd import _lib as L
    import ctypes as C
    lib = L.load()
    g = torch.Generator().manual_seed(9)
    B, n, K, nout = 5, 10, 5000, 256
    w = torch.randn(nout, K, generator=g)
    b = torch.randn(nout, generator=g)
    nullproj = torch.randn(nout, generator=g)
    cond = torch.zeros(B, K, dtype=torch.int64)
    cond[0, 4999] = 1                                   # one-hot (last column)
    cond[1, 0] = 1                                      # one-hot (first column)
    cond[2, torch.randint(0, K, (300,), generator=g)] = 1          # multi-hot
    cond[3] = torch.randint(-2, 3, (K,), generator=g)   # dense integer row: every list slot in use
    mask = torch.zeros(n, dtype=torch.uint8)            # row 4 all zero
    mask[7] = 1
    out = torch.full((n, nout), float("nan"), device="cuda")
    p = lambda t_: C.c_void_p(t_.data_ptr())
    wd, bd, nd, cd, md = w.cuda(), b.cuda(), nullproj.cuda(), cond.cuda(), mask.cuda()
    L.check(lib.sgd_linear_sparse_rows(p(cd), p(md), p(wd), p(bd), p(nd), B, n, nout, K, p(out), nout,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sparse_rows")
    got = out.cpu()
    ref = (cond.double() @ w.double().t() + b.double()).float().repeat(2, 1)
    ref[7] = nullproj
    assert torch.equal(got[0], w[:, 4999] + b) and torch.equal(got[1], w[:, 0] + b) and torch.equal(got[5], got[0])
    assert torch.equal(got[4], b) and torch.equal(got[7], nullproj)
    # (row 3 -- 5,000 non-zero entries -- is one sequential fp32 sum per output: ~sqrt(5000) roundings)
    err_sparse, err_dense = max_rel(got[[0, 1, 2, 4]], ref[[0, 1, 2, 4]]), max_rel(got, ref)
    assert err_sparse < 2e-6 and err_dense < 2e-5, (err_sparse, err_dense)
    # rows too long for the LDS list are refused (the caller keeps the dense kernel)
    assert lib.sgd_linear_sparse_rows(p(cd), p(md), p(wd), p(bd), p(nd), B, n, nout, 9000, p(out), nout, None) != 0


def test_unetca_label_map_and_nhot_equal_expanded():
    """stegoclusterlayout: uint8 label map + device n-hot == the [B,27,H,W] one-hot mask + CPU n-hot"""
    from sgdm_amd import guidance as G
    m, entry = build_model("ca_stego_c32_s16", "f16x3")
    g = torch.Generator().manual_seed(6)
    lm = torch.randint(0, 27, (2, 4, 4), generator=g).repeat_interleave(4, 1).repeat_interleave(4, 2).to(torch.uint8)
    x, t = torch.randn(2, 3, 16, 16, generator=g).cuda(), torch.tensor([500, 37]).cuda()
    with torch.no_grad():
        a = m.forward_with_cond_scale(x, t, cond_scale=2.0, cond=_ref_nhot(lm, 27).cuda(), layout=_ref_onehot_mask(lm, 27).cuda())
        b = m.forward_with_cond_scale(x, t, cond_scale=2.0, cond=G.stego_attr(lm.cuda(), 27), layout=lm.cuda())
    assert torch.equal(a, b)


def test_unet_fast_box_corners_equal_box_mask():
    """clusterlayout (LOST boxes): int box corners == rasterised [B,1,H,W] mask; ids == one-hot"""
    m, entry = build_model("uf_clusterlayout_c32_s16", "f32")
    k = entry["ctor"]["cond_dim"]
    g = torch.Generator().manual_seed(7)
    boxes = torch.tensor([[2, 3, 12, 15], [0, 5, 16, 9]])
    mask = torch.zeros(2, 1, 16, 16)
    for i, b in enumerate(boxes.tolist()):
        mask[i, 0, b[1]:b[3], b[0]:b[2]] = 1
    ids = torch.randint(0, k, (2,), generator=g)
    x, t = torch.randn(2, 3, 16, 16, generator=g).cuda(), torch.tensor([10, 900]).cuda()
    with torch.no_grad():
        a = m.forward_with_cond_scale(x, t, cond_scale=1.5, cond=F.one_hot(ids, k).float().cuda(), layout=mask.cuda())
        b = m.forward_with_cond_scale(x, t, cond_scale=1.5, cond=ids.cuda(), layout=boxes.cuda())
    assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------------------------
# against expansions produced by the reference's OWN functions / transform classes (tests/golden/vis.npz,
# make_golden_vis.py imports dataset/transforms/complex_ds_common_util.py)
# ------------------------------------------------------------------------------------------------------------------
def test_expansions_equal_reference_generated_fixtures():
    from conftest import load_npz
    from sgdm_amd import guidance as G
    v = load_npz("vis.npz")
    lm = torch.from_numpy(v["guid.labelmap"])
    assert torch.equal(G.onehot_layout(lm.cuda(), 27).cpu(), torch.from_numpy(v["guid.onehot"]))
    lm0 = lm.clone()
    lm0[lm0 == 255] = 0
    assert torch.equal(G.stego_attr(lm0.cuda(), 27).cpu(), torch.from_numpy(v["guid.nhot"]))
    # LOST boxes: corners mapped through the reference pipeline's index arithmetic, rasterised on the device, against the
    # mask the reference's RandomScaleCrop produced from the box drawn at the original size
    W0, H0 = (int(t) for t in v["guid.box_orig_size"])
    base, S = (int(t) for t in v["guid.box_crop_resize"])
    corners = [G.lost_box_in_frame([int(t) for t in b], (W0, H0), (int(p[0]), int(p[1])), (int(p[2]), int(p[3])), base, S)
               for b, p in zip(v["guid.box_orig"], v["guid.box_scaled_size_crop_xy"])]
    got = G.box_layout(torch.tensor(corners).cuda(), S, S).cpu()
    assert torch.equal(got, torch.from_numpy(v["guid.box_mask64"]))
