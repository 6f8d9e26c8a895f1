"""The persistent conv kernel when the device is NOT its own.  GPU only.

`sgd_igemm` runs one 512-thread block per compute unit with all of the CU's registers and a static tile list per block.
During the data-parallel backward RCCL's kernels hold CUs on a side stream (sgdm_amd/ddp.py): blocks that find no CU
start only when others exit.  Single-GPU stand-in for that: `sgd_debug_occupy` pins N compute units (512 threads +
150 KB of LDS per block: nothing fits beside it) on a side stream for a wall-clock interval while the conv launches run.

  * no hang, results bit-identical with and without the contention (a launch, the whole sampling evaluation, a
    training step);
  * with `grid_cap` = CUs - reserved (what training uses when world > 1) the slowdown under contention stays at the
    CU share taken; with the whole-device grid it is up to 2x -- both printed, the first asserted;
  * the balanced tail's bounded poll: stale arrival counters poison the output with NaN and set the health word of the
    workspace instead of hanging the device;
  * `bench.py --gpus 2` as a fresh child process (two ranks over gloo sharing the one GPU): the multi-GPU leg's rank
    plumbing, gradient exchange, initial-state broadcast and cross-rank checksum run end to end.
"""
import ctypes as C
import json
import math
import os
import subprocess
import sys
import time

import pytest
import torch

from test_hip_kernels import _lib, _p, _stream

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OCC = 32            # compute units pinned by the stand-in (4 per XCD under round-robin placement)


def _cus():
    return torch.cuda.get_device_properties(0).multi_processor_count


def _conv_args(L, lib, prec, n, cin, cout, h, work, grid_cap, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, h, cin, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).cuda()
    buf = torch.empty(lib.sgd_packed_weight_bytes(cout, cin, 3, prec) // 4, device="cuda")
    cp, op = C.c_int32(), C.c_int32()
    L.check(lib.sgd_pack_weight(_p(w), _p(buf), cout, cin, 3, prec, C.byref(cp), C.byref(op), _stream()), "pack")
    y = torch.full((n, h, h, cout), float("nan"), device="cuda")
    a = L.IgemmArgs()
    a.x0, a.c0 = x.data_ptr(), cin
    a.mode, a.n, a.hi, a.wi, a.ho, a.wo, a.stride = L.MODE_CONV3, n, h, h, h, h, 1
    a.w, a.cin_p, a.cout_p = buf.data_ptr(), cp.value, op.value
    a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cout, cout, prec
    if work is not None:
        a.work, a.work_bytes = work.data_ptr(), work.numel() * 4
    a.grid_cap = grid_cap
    return a, y, (x, w, buf)


def _timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def _occupy(lib, side, blocks, ms):
    """start the stand-in on the side stream and give it time to become resident (the stand-in lives in the diagnostics
    library, include/sgdm_hip_tools.h -- the product library `lib` does not export it)"""
    from sgdm_amd import _lib as L
    with torch.cuda.stream(side):
        assert L.load_tools().sgd_debug_occupy(blocks, float(ms), side.cuda_stream) == 0
    time.sleep(0.003)


@pytest.mark.parametrize("shape", [(256, 256, 32), (128, 128, 64)], ids=["c256_32", "c128_64"])
def test_conv_launch_under_cu_contention(shape):
    """one production 3x3 layer at UNet batch 80: whole-device grid vs the reserved grid, alone and next to the stand-in"""
    cin, cout, h = shape
    L, lib = _lib()
    prec = L.PREC_F16X3
    work = torch.zeros(int(lib.sgd_igemm_work_bytes()) // 4, device="cuda")
    side = torch.cuda.Stream()
    cus = _cus()
    res = {}
    outs = {}
    for name, cap in (("full", 0), ("reserved", cus - OCC)):
        a, y, keep = _conv_args(L, lib, prec, 80, cin, cout, h, work, cap)
        run = lambda: L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
        run()
        torch.cuda.synchronize()
        alone = min(_timed(run, 20) for _ in range(3))
        ref = y.clone()
        assert torch.isfinite(ref).all()
        y.fill_(float("nan"))
        _occupy(lib, side, OCC, 60.0)
        busy = _timed(run, 20)
        torch.cuda.synchronize()
        assert torch.equal(y, ref), name                 # bit-identical under contention
        res[name] = (alone, busy)
        outs[name] = ref
    # the two grids differ only in the summation order of the balanced tail's split tiles
    assert float((outs["full"] - outs["reserved"]).abs().max()) <= 2e-5 * float(outs["full"].abs().max())
    full_alone = res["full"][0]
    msg = (f"{cin}->{cout} @{h}^2 bs80: whole-device grid {res['full'][0]:.3f} ms alone, {res['full'][1]:.3f} ms next to "
           f"{OCC} busy CUs (x{res['full'][1] / full_alone:.2f}); grid_cap {cus - OCC}: {res['reserved'][0]:.3f} ms alone "
           f"(x{res['reserved'][0] / full_alone:.2f}), {res['reserved'][1]:.3f} ms next to them (x{res['reserved'][1] / full_alone:.2f})")
    print("\n" + msg)
    share = cus / float(cus - OCC)
    # done criterion: with the reserve the slowdown under contention is the CU share taken (+ measurement slack), and the
    # contention itself costs the reserved grid nothing
    assert res["reserved"][1] <= res["reserved"][0] * 1.08, msg
    assert res["reserved"][1] <= full_alone * share * 1.12, msg
    assert int(work.view(torch.int32)[int(lib.sgd_igemm_work_status_offset()) // 4]) == 0


def test_sampling_evaluation_under_cu_contention():
    """one CFG evaluation of C2 at UNet batch 80 (default: whole-device grid) next to the stand-in: no hang, same bits"""
    import bench
    L, lib = _lib()
    model, _, data = bench.build_model(bench.WORKLOADS["c2"], torch.device("cuda"), "f16x3", 40)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(40, 3, 64, 64, generator=g).cuda()
    t = torch.randint(0, 1000, (40,), generator=g).cuda()
    cond = data["cond"].cuda()
    side = torch.cuda.Stream()
    with torch.no_grad():
        run = lambda: model.forward_with_cond_scale(x, t, cond_scale=2.0, cond=cond)
        ref = run()
        torch.cuda.synchronize()
        alone = min(_timed(run, 3) for _ in range(2))
        _occupy(lib, side, OCC, 200.0)
        got = None

        def keep():
            nonlocal got
            got = run()
        busy = _timed(keep, 3)
        torch.cuda.synchronize()
    assert torch.equal(got, ref)
    print(f"\nC2 CFG evaluation at UNet batch 80: {alone:.2f} ms alone, {busy:.2f} ms next to {OCC} busy CUs "
          f"(x{busy / alone:.2f}; static tile lists on the whole-device grid: a launch whose blocks do not all fit takes up to 2x)")
    assert busy < 2.3 * alone


def test_training_step_with_reserve_under_cu_contention():
    """a training step with the reserve training uses when world > 1 -- on the BACKWARD program, which is what the exchange
    overlaps; the forward keeps the whole device: same gradients bit for bit with and without the stand-in on the side
    stream under the backward, and the contention costs (almost) nothing"""
    import bench
    from sgdm_amd.diffusion import LatentDiffusion
    from sgdm_amd.synth import synth_batch
    L, lib = _lib()
    wl = bench.WORKLOADS["c2"]
    B = 16
    model, _, _ = bench.build_model(wl, torch.device("cuda"), "f16x3", B)
    model.dropout = 0.0
    model.train()
    model.hip_reserve_cus = OCC
    diff = LatentDiffusion(device="cuda", **bench.MODEL_PARAMS).train()
    diff.set_denoise_fn(model.forward, model.forward_with_cond_scale)
    data = synth_batch(wl["method"], B, 64, wl["cond_dim"], 0, seed=5)
    g = torch.Generator().manual_seed(5)
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    noise = torch.randn(B, 3, 64, 64, generator=g).cuda()
    mask = (torch.rand(B, generator=g) < 0.1).cuda()
    x0, cond = data["image"].cuda(), data["cond"].cuda()
    side = torch.cuda.Stream()

    def step(occupy_ms=0.0):
        for p in model.parameters():
            p.grad = None
        loss, _ = diff.p_losses(x0, t, noise, cond=cond, cond_drop_prob=0.1, cond_drop_mask=mask)
        if occupy_ms:                # where a bucket's all-reduce starts: behind the forward, under the backward
            side.wait_stream(torch.cuda.current_stream())
            _occupy(lib, side, OCC, occupy_ms)
        loss.backward()
        return loss

    step()
    torch.cuda.synchronize()
    eng = next(iter(model._engines.values()))
    assert eng._grid_cap == _cus() - OCC
    assert all(a.grid_cap == 0 for a, _ in eng._late) and all(a.grid_cap == eng._grid_cap for a, _ in eng.backward.late)
    alone = min(_timed(step, 2) for _ in range(2))
    ref = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    # the stand-in holds its CUs for about half a step: it starts with the backward and is gone before the next forward
    busy = _timed(lambda: step(0.5 * alone), 2)
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, ref[k]), k
    print(f"\ntrain step bs{B} with {OCC} CUs reserved: {alone:.2f} ms alone, {busy:.2f} ms next to {OCC} busy CUs (x{busy / alone:.2f})")
    assert busy < 1.15 * alone
    # sampling on the same model goes back to the whole device
    model.eval()
    with torch.no_grad():
        model.forward_with_cond_scale(x0, t, cond_scale=2.0, cond=cond)
    eng2 = model._engines[(2 * B, 64, 64, L.PREC_F16X3)]
    assert eng2._grid_cap == 0 if hasattr(eng2, "_grid_cap") else True


def test_balanced_tail_poll_is_bounded():
    """stale arrival counters (what a faulted launch or a second stream on the same workspace would leave): the finisher
    gives up after its bounded poll, poisons ITS tile with NaN and raises the health word -- the device does not hang"""
    L, lib = _lib()
    # 1288 tiles of 8 chunks on 256 blocks: 5 whole rounds + 1 tile per XCD, split 4 ways over the idle blocks
    n, cin, cout, h = 161, 256, 128, 32
    work = torch.zeros(int(lib.sgd_igemm_work_bytes()) // 4, device="cuda")
    a, y, keep = _conv_args(L, lib, L.PREC_F16X3, n, cin, cout, h, work, 0)
    lay = (C.c_int32 * (4 * 256))()
    assert lib.sgd_igemm_tail_layout(n * 8, 8, 9, 256, lay) == 0
    split_blocks = [b for b in range(256) if lay[4 * b] > 0]
    assert split_blocks, "geometry must exercise the balanced tail"
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
    torch.cuda.synchronize()
    ref = y.clone()
    assert torch.isfinite(ref).all()
    status = int(lib.sgd_igemm_work_status_offset()) // 4
    wi = work.view(torch.int32)
    assert int(wi[status]) == 0 and int(wi[:status].abs().sum()) == 0          # counters reset themselves
    # a counter that can never reach its target: a very negative arrival count on one split tile
    cnt = lay[4 * split_blocks[0] + 1]
    wi[2 * cnt] = -(1 << 30)
    y.fill_(0.0)
    t0 = time.time()
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert int(wi[status]) == 1, "health word not raised"
    bad = ~torch.isfinite(y)
    assert bad.any() and bad.sum() <= 128 * cout                                # exactly the one tile is poisoned
    assert torch.equal(y[~bad], ref[~bad])
    assert dt < 30.0
    print(f"\nbounded poll expired after {dt:.2f} s; {int(bad.sum())} poisoned outputs of {y.numel()}")
    # recovery: zero the workspace, run again
    work.zero_()
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
    torch.cuda.synchronize()
    assert torch.equal(y, ref) and int(wi[status]) == 0


def test_health_word_is_sticky_and_the_host_paths_raise():
    """a balanced-tail time-out used to leave counters that later launches could trip over
    silently.  Now (1) while the health word is up EVERY split tile on that workspace is poisoned (a launch with healthy
    counters next to a raised word returns NaN in its split tiles, nothing else), (2) the host paths that own a workspace look
    at the word: the sampler at the end of a trajectory (`_Engine.check_health`, synchronising) and the training step
    through its asynchronous copy (`note_health` / `poll_health`) -- both raise and re-zero the workspace."""
    import bench
    from sgdm_amd.diffusion import LatentDiffusion
    L, lib = _lib()
    # ---- (1) kernel: the geometry of test_balanced_tail_poll_is_bounded, counters clean, word raised by hand
    n, cin, cout, h = 161, 256, 128, 32
    work = torch.zeros(int(lib.sgd_igemm_work_bytes()) // 4, device="cuda")
    a, y, keep = _conv_args(L, lib, L.PREC_F16X3, n, cin, cout, h, work, 0)
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
    torch.cuda.synchronize()
    ref = y.clone()
    status = int(lib.sgd_igemm_work_status_offset()) // 4
    wi = work.view(torch.int32)
    wi[status] = 1
    y.fill_(0.0)
    t0 = time.time()
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
    torch.cuda.synchronize()
    assert time.time() - t0 < 5.0                                  # no polling for a time-out: the word alone decides
    bad = ~torch.isfinite(y)
    assert bad.any() and torch.equal(y[~bad], ref[~bad]) and int(wi[status]) == 1
    assert int(wi[:status].abs().sum()) == 0                        # the counters still reset themselves
    work.zero_()
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    # ---- (2) host: sampler and training step
    wl = bench.WORKLOADS["c2"]
    B = 4
    model, _, data = bench.build_model(wl, torch.device("cuda"), "f16x3", B)
    diff = LatentDiffusion(device="cuda", **bench.MODEL_PARAMS)
    diff.set_denoise_fn(model.forward, model.forward_with_cond_scale)
    skw = dict(sampling_method="ddim", num_timesteps=2, ddim_eta=0.0, log_num_per_prog=2, clip_denoised=True, dtp=1,
               temperature=1.0, noise_dropout=0, random_sample_condition=False, return_inter_dict=True)
    kw = dict(cond=data["cond"].cuda(), layout=None, cond_scale=2.0)
    with torch.no_grad():
        diff.p_sample_loop("ddim", (B, 3, 64, 64), skw, denoise_sample_fn_kwargs=dict(kw), condition_kwargs={})     # healthy: no raise
        eng = next(iter(model._engines.values()))
        assert eng.work_bytes > 0
        eng.work.view(torch.int32)[status] = 1
        with pytest.raises(RuntimeError, match="balanced tail timed out"):
            diff.p_sample_loop("ddim", (B, 3, 64, 64), skw, denoise_sample_fn_kwargs=dict(kw), condition_kwargs={})
        assert int(eng.work.view(torch.int32).abs().sum()) == 0      # re-zeroed
        diff.p_sample_loop("ddim", (B, 3, 64, 64), skw, denoise_sample_fn_kwargs=dict(kw), condition_kwargs={})     # and usable again
    model.train()
    model.dropout = 0.0
    diff.train()
    x0 = data["image"].cuda()
    loss, _ = diff.forward_tao(x0, cond=kw["cond"], cond_drop_prob=0.1)
    loss.backward()
    teng = model._engines[(B, 64, 64, L.PREC_F16X3)]
    torch.cuda.synchronize()
    teng.poll_health()                                              # clean step: nothing
    teng.work.view(torch.int32)[status] = 1
    teng.note_health()                                              # (what Backward.run does after its program)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="balanced tail timed out"):
        diff.forward_tao(x0, cond=kw["cond"], cond_drop_prob=0.1)    # the next step's prepare() looks at the copy
    assert int(teng.work.view(torch.int32).abs().sum()) == 0
    loss, _ = diff.forward_tao(x0, cond=kw["cond"], cond_drop_prob=0.1)
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # ---- (3) ADVICE round 5 (medium): the poisoned step must not be APPLIED.  The backward derives a flag from the word, the
    # fused optimizer takes it as skip_if_nonzero (device side, no host sync), the next step's prepare() raises and clears it.
    from sgdm_amd.optim import FusedAdamWEma
    from sgdm_amd.unet import grad_health
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdamWEma(params, lr=1e-3)
    opt.step()                                                      # healthy step: parameters move
    torch.cuda.synchronize()
    before = [p.detach().clone() for p in params]
    moments = [opt.state[p]["exp_avg"].clone() for p in params if opt.state[p]]
    opt.zero_grad(set_to_none=True)
    teng.work.view(torch.int32)[status] = 1                         # a launch of this iteration timed out
    loss, _ = diff.forward_tao(x0, cond=kw["cond"], cond_drop_prob=0.1)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert float(grad_health("cuda")[0]) > 0
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, params))          # nothing was written
    assert all(torch.equal(a, opt.state[p]["exp_avg"]) for a, p in zip(moments, [p for p in params if opt.state[p]]))
    with pytest.raises(RuntimeError, match="repeat the iteration"):
        diff.forward_tao(x0, cond=kw["cond"], cond_drop_prob=0.1)
    assert float(grad_health("cuda")[0]) == 0 and int(teng.work.view(torch.int32).abs().sum()) == 0
    opt.zero_grad(set_to_none=True)
    loss, _ = diff.forward_tao(x0, cond=kw["cond"], cond_drop_prob=0.1)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in params)
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, params))


def test_bench_two_ranks_gloo_on_one_gpu():
    """`python bench.py --gpus 2 ...` as a fresh child process: it launches its two ranks itself (children spawned before
    the GPU is touched), the ranks share the one GPU over gloo (SGDM_DIST_BACKEND), and rank 0's line must show a
    2-rank job: both halves of the metric, per-rank timings and equal gradient checksums after the exchange"""
    env = dict(os.environ, SGDM_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    assert line["value"] > 0 and line["config"]["batch_per_gpu"] == 40
    tr = line["train_step"]
    assert tr["world_size"] == 2 and tr["backend"] == "gloo" and tr["global_batch"] == 160
    assert len(tr["per_rank_ms"]) == 2 and all(v > 0 for v in tr["per_rank_ms"])
    assert len(tr["grad_checksum_first_step"]) == 2 and tr["grad_checksums_equal"] is True
    assert tr["reserved_cus"] == 0                  # gloo launches no kernels on the device: nothing to reserve
    # the overlap record of the exchange (host-ordered under gloo, HIP events all the same)
    assert tr["exchange"] is True and tr["exchange_ms"] is not None and tr["exposed_exchange_ms"] is not None
    assert 0.0 < tr["first_bucket_at_frac_of_backward"] < 1.0
    assert len(tr["exchange_buckets"]) >= 3
    assert line["train_step_bs40"]["batch_per_gpu"] == 40 and line["train_step_bs40"]["global_batch"] == 80
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None
