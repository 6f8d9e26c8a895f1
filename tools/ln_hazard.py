#!/usr/bin/env python3
"""(needs the diagnostic code of profiles/r5_igemm_experiments.patch applied to csrc/igemm.hip: `git apply profiles/r5_igemm_experiments.patch`)
LayerNorm-row prologue on the two-plane flat instance: how often it goes wrong, where, and what
makes it stop.  Needs a library built with -DSGDM_FLAT2_LN -DSGDM_EXP:
    SGDM_BUILD_TAG=_exp SGDM_EXTRA_FLAGS="-DSGDM_FLAT2_LN -DSGDM_EXP" python self-guided-diffusion-models_amd/build.py
    SGDM_LIB_PATH=.../libsgdm_hip_exp.so python tools/ln_hazard.py [--prec f16x3] [--reps 20]
One launch shape (Attention_LR to_q at UNet batch 160: m 40960, 512 -> 512).  Reference = the one-plane instance of the same
library (SGDM_FLAT2=0); every SGDM_EXP mask (csrc/igemm.hip: EXP_HOOK) is then run `reps` times on the two-plane instance."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=40960); ap.add_argument("--cin", type=int, default=512)
ap.add_argument("--cout", type=int, default=512); ap.add_argument("--prec", default="f16x3")
ap.add_argument("--reps", type=int, default=20); ap.add_argument("--masks", default="0,1,8,9,4,2")
ap.add_argument("--nobeta", action="store_true")
ap.add_argument("--identity", action="store_true", help="W = I: the output IS the staged input; say what the wrong rows hold")
ap.add_argument("--select", type=int, default=-1, help="with --identity and cout < cin: W picks channels select .. select + cout (ONE N tile: one block per row tile)")
ap.add_argument("--side", action="store_true", help="with --identity and mask bit 128: compare the loader's side copy of element 0 (before the split / LDS store) with what the MFMA saw")
a = ap.parse_args()
lib = L.load(); prec = L.PREC_BY_NAME[a.prec]
dev = "cuda"
torch.manual_seed(0)
m, cin, cout = a.m, a.cin, a.cout
x = torch.randn(m, cin, device=dev)
w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
if a.identity and a.select >= 0:
    w = torch.zeros(cout, cin, device=dev)
    w[torch.arange(cout), a.select + torch.arange(cout)] = 1.0
    w = w.reshape(cout, cin, 1, 1).contiguous()
elif a.identity:
    assert cin == cout
    w = torch.eye(cin, device=dev).reshape(cout, cin, 1, 1).contiguous()
gamma, beta = torch.randn(cin, device=dev), torch.randn(cin, device=dev)
stats = torch.empty(m, 2, device=dev)
st = torch.cuda.current_stream().cuda_stream
L.check(lib.sgd_ln_stats(C.c_void_p(x.data_ptr()), m, cin, C.c_float(1e-5), C.c_void_p(stats.data_ptr()), st), "ln_stats")
buf = torch.empty(lib.sgd_packed_weight_bytes(cout, cin, 1, prec) // 4, device=dev)
cp, op = C.c_int32(), C.c_int32()
L.check(lib.sgd_pack_weight(C.c_void_p(w.data_ptr()), C.c_void_p(buf.data_ptr()), cout, cin, 1, prec, C.byref(cp), C.byref(op), st), "pack")
y = torch.empty(m, cout, device=dev)
g = L.IgemmArgs()
g.x0, g.c0 = x.data_ptr(), cin
g.mode, g.m, g.rows_per_n, g.stride = L.MODE_FLAT, m, m, 1
g.pro, g.pa, g.pb = L.PRO_LN_ROW, stats.data_ptr(), gamma.data_ptr()
if not a.nobeta: g.pc = beta.data_ptr()
g.w, g.cin_p, g.cout_p = buf.data_ptr(), cp.value, op.value
side_all = torch.full((m, cin // 4), float("nan"), device=dev)
if a.side:
    g.x1 = side_all.data_ptr()                      # (c1 stays 0: the kernel does not read it; -DSGDM_EXP bit 128 writes it)
g.y, g.cout, g.y_ld, g.prec = y.data_ptr(), cout, cout, prec


def run(flat2, mask):
    os.environ["SGDM_FLAT2"] = str(flat2); os.environ["SGDM_EXP"] = str(mask)
    y.fill_(float("nan"))
    L.check(lib.sgd_igemm(C.byref(g), st), "igemm")
    torch.cuda.synchronize()
    return y.clone()


def explain(out, ref):
    """W = I: for wrong (row, 32-channel plane) cells, look for the values they hold among the cells of the same 128-row tile"""
    P = cin // 32
    o, r = out.view(m, P, 32), ref.view(m, P, 32)
    badc = (o != r).any(dim=2)                                  # [m, P]
    cells = badc.nonzero()
    print(f"  wrong cells {len(cells)} in {int(badc.any(dim=1).sum())} rows; planes hit (count per plane): {badc.sum(dim=0).tolist()}")
    per_row = badc.sum(dim=1)
    print(f"  wrong planes per wrong row: min {int(per_row[per_row > 0].min())} max {int(per_row.max())}")
    # a loader thread stages rows arow + 32 j (j = 0..3) of a tile: do the four items of a thread fail together?
    tb = badc.view(m // 128, 4, 32, P)                          # [tile, j, arow, plane]
    per_thread = tb.sum(dim=1)                                  # wrong items per (tile, arow, plane)
    print(f"  wrong items per (tile, thread row, plane) that has any: {torch.bincount(per_thread[per_thread > 0].flatten(), minlength=5).tolist()[1:]} (1, 2, 3, 4 of 4)")
    print(f"  wrong cells per item index j: {tb.sum(dim=(0, 2, 3)).tolist()}")
    # thread row lt >> 3 = 8 * (loader wave) + row of the wave: which of the four loader waves (one per SIMD) wrote them
    print(f"  wrong cells per loader wave (SIMD): {tb.sum(dim=(0, 1, 3)).view(4, 8).sum(dim=1).tolist()}")
    el = (o != r).view(m, P, 8, 4)                              # [row, plane, quad, element]
    print(f"  wrong values per element of the quad: {el.sum(dim=(0, 1, 2)).tolist()}; per quad of the row: {el.sum(dim=(0, 1, 3)).tolist()}")
    kinds = {}
    for row, pl in cells[:: max(1, len(cells) // 200)][:200].tolist():
        v = o[row, pl]
        t0 = row // 128 * 128
        cand = r[t0:t0 + 128]                                   # [128, P, 32]
        hit = (cand == v).all(dim=2).nonzero()
        nz = int((v != r[row, pl]).sum())
        if len(hit):
            rr, pp = hit[0].tolist()
            key = f"holds row {rr - (row - t0):+d} plane {pp - pl:+d}"
        elif bool(torch.isnan(v).any()):
            key = "NaN"
        else:
            d = (v - r[row, pl]).abs().max().item()
            key = f"no match ({nz}/32 values differ)" if d > 1e-3 else f"small difference <=1e-3 ({nz}/32 values)"
        kinds[key] = kinds.get(key, 0) + 1
    for k, c in sorted(kinds.items(), key=lambda kv: -kv[1])[:12]:
        print(f"    {c:4d} x {k}")
    for row, pl in cells[:: max(1, len(cells) // 6)][:6].tolist():
        c = 32 * pl + int((o[row, pl] != r[row, pl]).nonzero()[0])
        mean, rstd = stats[row, 0].item(), stats[row, 1].item()
        t = (x[row, c].item() - mean) * rstd
        print(f"  row {row} channel {c}: got {out[row, c].item():.7g} ref {ref[row, c].item():.7g} | x {x[row, c].item():.6g} mean {mean:.6g} "
              f"rstd {rstd:.6g} t {t:.7g} t*gamma {t * gamma[c].item():.7g} t*gamma+beta {t * gamma[c].item() + beta[c].item():.7g} "
              f"gamma {gamma[c].item():.6g} beta {beta[c].item():.6g} | x-mean {x[row, c].item() - mean:.7g} x*rstd {x[row, c].item() * rstd:.7g}")
        for rr in (row - 1, row + 1, row ^ 1):
            if 0 <= rr < m:
                t2 = (x[row, c].item() - stats[rr, 0].item()) * stats[rr, 1].item()
                print(f"      with the statistics of row {rr}: t*gamma+beta {t2 * gamma[c].item() + beta[c].item():.7g}")
    row, pl = cells[0].tolist()
    print(f"  first wrong cell row {row} (row in tile {row % 128}) plane {pl}:\n    got {o[row, pl, :8].tolist()}\n    ref {r[row, pl, :8].tolist()}")


ref = run(0, 0)
again = run(0, 0)
print(f"m={m} cin={cin} cout={cout} prec={a.prec}: one-plane instance repeatable: {bool((ref == again).all())}")
for mask in [int(t) for t in a.masks.split(",")]:
    bad_launches, rows_total, hist = 0, 0, [0] * 8
    worst = 0.0
    for _ in range(a.reps):
        out = run(1, mask)
        bad = (out != ref).any(dim=1)
        nb = int(bad.sum())
        if nb:
            bad_launches += 1; rows_total += nb
            idx = bad.nonzero().flatten()
            for r8 in range(8): hist[r8] += int(((idx % 8) == r8).sum())
            worst = max(worst, float((out - ref).abs().max()))
            if a.identity and a.select < 0 and bad_launches == 1: explain(out, ref)
            if a.identity and a.side and bad_launches == 1:
                torch.cuda.synchronize()
                sel0 = max(a.select, 0)
                side = side_all[:, sel0 // 4: sel0 // 4 + cout // 4]     # the quads W shows (one N tile: written by the same block)
                beta_v = beta[sel0: sel0 + cout]
                e0_out, e0_ref = out.view(m, cout // 4, 4)[..., 0], ref.view(m, cout // 4, 4)[..., 0]
                wrong = e0_out != e0_ref                              # element 0 of a quad as the MFMA saw it
                written = ~torch.isnan(side)
                if mask & 768:        # the side buffer holds an intermediate: x - mean (512) or (x - mean) * rstd (256)
                    xs = x[:, sel0: sel0 + cout].view(m, cout // 4, 4)[..., 0]
                    inter = xs - stats[:, 0:1]
                    if mask & 256: inter = inter * stats[:, 1:2]
                    bad_i = (side - inter).abs() > 1e-5 * inter.abs().clamp_min(1e-2)
                    print(f"  intermediate {'x - mean' if mask & 512 else '(x - mean) * rstd'} of element 0: wrong at the MFMA {int(wrong.sum())} quads; "
                          f"of those the intermediate is wrong too: {int((wrong & bad_i).sum())}, right: {int((wrong & ~bad_i).sum())}; "
                          f"intermediate wrong elsewhere: {int((~wrong & bad_i).sum())}")
                    for r_, q_ in wrong.nonzero()[:6].tolist():
                        print(f"    row {r_} channel {sel0 + 4 * q_}: intermediate {side[r_, q_].item():.7g} expected {inter[r_, q_].item():.7g} | "
                              f"MFMA {e0_out[r_, q_].item():.7g} expected {e0_ref[r_, q_].item():.7g} beta {beta_v[4 * q_].item():.7g}")
                    continue
                sw = (side - e0_ref).abs() > 1e-3 * e0_ref.abs().clamp_min(1e-3)     # side copy is fp32, the output hi + lo
                print(f"  (each row tile staged by {(cout + 127) // 128} block(s)) side copy written for {int(written.sum())} of {side.numel()} quads; element 0 wrong at the MFMA: {int(wrong.sum())}; "
                      f"of those the loader's own fp32 copy is ALSO wrong: {int((wrong & sw).sum())}, right: {int((wrong & ~sw & written).sum())}; "
                      f"side copy wrong where the MFMA saw the right value: {int((~wrong & sw & written).sum())}")
                for title, sel in (("MFMA wrong, side copy right", wrong & ~sw & written), ("both wrong", wrong & sw),
                                   ("side copy wrong, MFMA right", ~wrong & sw & written)):
                    for r_, q_ in sel.nonzero()[:5].tolist():
                        c_ = 4 * q_
                        e0 = e0_ref[r_, q_].item()
                        print(f"    {title}: row {r_} channel {sel0 + c_}: side {side[r_, q_].item():.7g}  MFMA {e0_out[r_, q_].item():.7g}  "
                              f"expected {e0:.7g}  beta {beta_v[c_].item():.7g}  lo(expected) {e0 - float(torch.tensor(e0).half()):.3g}")
    print(f"SGDM_EXP={mask:2d}: {bad_launches}/{a.reps} launches differ, {rows_total} rows, rows mod 8 {hist}, max|d| {worst:.3g}")
