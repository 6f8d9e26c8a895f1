"""Guidance-tensor ingest on the device.

The reference builds the UNet's guidance tensors per sample on CPU data workers and ships them over PCIe
(dataset/ds_utils/unsupervised_cluster.py:33-46, dataset/transforms/complex_ds_common_util.py:103-162,243-253):

    cluster / label  : F.one_hot(id, k) int64            [B, k]         (k = 5000: 40 KB per sample for ONE integer)
    stego masks      : one-hot over 27 classes, float    [B, 27, 64, 64] (442 KB per sample for a 4 KB label map)
    stego attributes : n-hot of the labels present       [B, 27]
    LOST boxes       : rasterised box mask, float        [B, 1, 64, 64] (16 KB per sample for four integers)

Here the COMPACT forms travel and the expansion happens inside the boundary kernels of the UNet evaluation:

    model(x, t, cond=ids,            ...)   ids    int64 [B]            -> one-hot rows inside sgd_cond_select; for unet_fast
                                                                           with k >= 1024 mlp_cond.0 becomes a weight-column
                                                                           gather (sgd_linear_gather), bit-identical
    model(x, t, layout=label_map,    ...)   uint8 [B,H,W] or [B,1,H,W]  -> one-hot channels inside sgd_pack_input_compact
    model(x, t, layout=boxes,        ...)   int32/int64 [B,4] (x0,y0,x1,y1 in the H x W frame) -> box mask, same kernel
    stego_attr(label_map, k)                                             -> n-hot float [B,k] (sgd_labelmap_nhot)

The expanded (reference) forms keep working unchanged.  The functions below are the stand-alone device expansions,
bit-exact against the reference formulas (tests/test_hip_guidance.py); they need the GPU, there is no CPU fallback.
"""
import ctypes as C

import torch

from . import _lib as L


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(t):
    if t.device.type != "cuda":
        raise RuntimeError("sgdm_amd.guidance runs on the MI355X HIP path only (inputs must be on a cuda device)")


def stego_attr(label_map, k):
    """stegomask_to_attr_nhot (complex_ds_common_util.py:126-133): uint8 label map [B,H,W] -> float n-hot [B,k]"""
    _need_cuda(label_map)
    lm = label_map.reshape(label_map.shape[0], -1).contiguous()
    assert lm.dtype == torch.uint8
    out = torch.empty(lm.shape[0], k, device=lm.device, dtype=torch.float32)
    L.check(L.load().sgd_labelmap_nhot(_p(lm), lm.shape[0], lm.shape[1], k, _p(out), _stream()), "sgd_labelmap_nhot")
    return out


def onehot_layout(label_map, num_classes):
    """stego_to_onehotmask (complex_ds_common_util.py:118-123): uint8 label map [B,H,W] -> float [B,num_classes,H,W]"""
    _need_cuda(label_map)
    B, H, W = label_map.shape[0], label_map.shape[-2], label_map.shape[-1]
    lm = label_map.reshape(B, H, W).contiguous()
    x = torch.zeros(B, 1, H, W, device=lm.device)
    nhwc = torch.empty(B, H, W, 1 + num_classes, device=lm.device)
    lib = L.load()
    L.check(lib.sgd_pack_input_compact(_p(x), _p(lm), 1, None, None, B, B, 1, num_classes, H, W, _p(nhwc), _stream()),
            "sgd_pack_input_compact")
    return nhwc[..., 1:].permute(0, 3, 1, 2).contiguous()


def box_layout(boxes, H, W):
    """get_lostbboxmask (complex_ds_common_util.py:151-162): int (x0,y0,x1,y1) [B,4] -> float {0,1} mask [B,1,H,W]"""
    _need_cuda(boxes)
    B = boxes.shape[0]
    bx = boxes.to(torch.int32).contiguous()
    x = torch.zeros(B, 1, H, W, device=bx.device)
    nhwc = torch.empty(B, H, W, 2, device=bx.device)
    L.check(L.load().sgd_pack_input_compact(_p(x), _p(bx), 2, None, None, B, B, 1, 1, H, W, _p(nhwc), _stream()),
            "sgd_pack_input_compact")
    return nhwc[..., 1:].permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------------------------
# box corners through the reference's data pipeline (host integer bookkeeping, bit-exact)
# ------------------------------------------------------------------------------------------------------------------
def _nearest_range(a, b, src, dst):
    """[a, b) on a length-`src` axis -> the index range [lo, hi) of a PIL NEAREST resize to `dst` that samples inside it:
    output j reads input int((j + 0.5) * src / dst) (Pillow's nearest filter: centre of the output pixel, truncated)"""
    scale = src / dst
    hit = [j for j in range(dst) if a <= int((j + 0.5) * scale) < b]
    return (hit[0], hit[-1] + 1) if hit else (0, 0)


def lost_box_in_frame(box, orig_size, scaled_size, crop_xy, crop_size, out_size):
    """The reference rasterises a LOST box at the ORIGINAL image size (get_lostbboxmask, complex_ds_common_util.py:
    151-162) and pushes the MASK through the image's own transform (RandomScaleCrop, :16-99): PIL NEAREST resize to
    `scaled_size` = (ow, oh), crop of `crop_size` at `crop_xy` = (x1, y1), PIL NEAREST resize to `out_size`.  A nearest
    resampling of an axis-aligned box is an axis-aligned box: this returns ITS corners (x0, y0, x1, y1) in the
    out_size x out_size frame -- the contract of ``box_layout`` / of int box corners passed as `layout`: corners are given
    in the frame the UNet sees, produced by this index mapping, not by scaling the original corners arithmetically.
    Pinned against masks produced by the reference's own classes (tests/golden/vis.npz, tests/test_hip_guidance.py)."""
    (W0, H0), (ow, oh), (cx, cy) = orig_size, scaled_size, crop_xy
    out = []
    for (a, b, src, mid, c0) in ((box[0], box[2], W0, ow, cx), (box[1], box[3], H0, oh, cy)):
        lo, hi = _nearest_range(a, b, src, mid)                     # after the random-scale resize
        lo, hi = max(lo - c0, 0), min(hi - c0, crop_size)           # after the crop
        lo, hi = _nearest_range(lo, hi, crop_size, out_size) if hi > lo else (0, 0)
        out.append((lo, hi))
    return [out[0][0], out[1][0], out[0][1], out[1][1]]
