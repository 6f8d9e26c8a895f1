#!/usr/bin/env python3
"""Benchmark of the hot path: classifier-free-guided 1000-step DDPM sampling of 64x64 images.

    python bench.py --gpus N --steps K --warmup W [--prec f32|f16x3|bf16x3] [--workload c2|c5]

A "step" is one CFG sampling step of the reference's native sampler on one batch of synthetic input:
one UNet evaluation at 2B (cond + uncond) + the fused x0/posterior update (ddpm_sampler.py:154-192).
Workload (BASELINE.json configs[1]): ImageNet-64 `unet_fast` ch=128, self-labeled cluster k=5000,
cond_scale=2, bs=40 per GPU (UNet batch 80); random-init weights, synthetic inputs (no datasets offline).
Sampling shards by images with no collective: N ranks sample N independent batches (weak scaling).

metric value = images/s of a full 1000-step trajectory = (N * B) / (1000 * t_step).

One JSON line on rank 0 with `roofline` (dominant kernel: the fused implicit-GEMM conv, timed with HIP
events around every launch in an instrumented pass) and `cpu_baseline` (the CPU oracle -- the
restatement of the reference's UNet -- timed on this box's host cores on a bounded sample), plus sub-records of
the same run: `train_step` (metric's second half; `train_step_bs40`: configs[2]'s per-GPU batch; `exchange_world1`: the
data-parallel step through a one-rank RCCL group), `f32_exact` (the headline workload in exact-fp32 MFMA arithmetic), `c5` /
`c4` (BASELINE.json configs[4] / [3], unetca_fast bs=80: sampling step + train step) and `c1` (configs[0] at its true shape
on the GPU, eager launches vs the hipGraph-captured step).  `roofline` also carries the device's own ceiling measured in the
run (`device_mfma_tflops`, `frac_of_device_ceiling`, `device_copy_tbps`).

Multi-GPU: the driver launches one rank per GPU through torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment).  Run by hand as `python bench.py --gpus N` it launches the N ranks ITSELF -- as child processes, before
this process has touched the GPU -- over RCCL on 127.0.0.1 and relays rank 0's JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "self-guided-diffusion-models_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


class AD(dict):
    __getattr__ = dict.__getitem__


WORKLOADS = {
    # BASELINE.json configs[1] (C2): IN64 unet_fast ch128 cluster k=5000 w=2 bs=40
    "c2": dict(kind="unet_fast", batch=40, image=64, cond_dim=5000, method="cluster", layout_dim=0,
               desc="IN64 unet_fast ch128 cluster-k5000 cond_scale=2 bs=40/GPU (UNet batch 80), 1000-step native DDPM",
               gflop_per_eval_img=79.27, mb_per_eval_img=192.2, weights_mb=301.1),
    # BASELINE.json configs[4] (C5): COCO-Stuff-64 unetca_fast stegoclusterlayout L=27 bs=80
    "c5": dict(kind="unetca_fast", batch=80, image=64, cond_dim=27, method="stegoclusterlayout", layout_dim=27,
               desc="COCO-Stuff64 unetca_fast stegoclusterlayout L=27 cond_scale=2 bs=80/GPU (UNet batch 160), 1000-step native DDPM",
               gflop_per_eval_img=67.89, mb_per_eval_img=165.8, weights_mb=253.1),
    # BASELINE.json configs[3] (C4): VOC-64 unetca_fast self-boxed clusterlayout (LOST) cond_dim=100 context_dim=32 bs=80
    "c4": dict(kind="unetca_fast", batch=80, image=64, cond_dim=100, method="clusterlayout", layout_dim=1,
               desc="VOC64 unetca_fast clusterlayout(LOST) cond_dim=100 ctx=32 cond_scale=2 bs=80/GPU (UNet batch 160), 1000-step native DDPM",
               gflop_per_eval_img=67.65, mb_per_eval_img=165.4, weights_mb=253.3),
}
MODEL_PARAMS = dict(given_betas=None, beta_schedule="linear", linear_start=0.0001, linear_end=0.02, cosine_s=8e-3,
                    v_posterior=0.0, logvar_init=0.0, learn_logvar=False, clip_denoised=True,
                    parameterization="eps", log_num_per_prog=10, loss_type="l2", sampling="native",
                    num_timesteps=1000)
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0 / 3, "bf16x3": 2500.0 / 3}     # MI355X_MICROARCH.md chip table
# BASELINE.json configs[0] (C1): cifar10 unet_fast ch=64 label K=10, 32x32, 10-step DDIM eta=0, w=2, bs=8
C1 = dict(kind="unet_fast", batch=8, image=32, cond_dim=10, method="label", layout_dim=0, model_channels=64, ddim_steps=10,
          desc="cifar10 unet_fast ch64 label-k10 32x32 cond_scale=2 bs=8, 10-step DDIM eta=0", gflop_per_eval_img=4.94)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (this process has not initialised
    the GPU -- device_count() does not) and relay rank 0's stdout; exit code = worst child."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # poll ALL children: if any rank dies, the others would sit in RCCL until its timeout -- end them instead
    import threading
    out_box = []
    reader = threading.Thread(target=lambda: out_box.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write(out_box[0] if out_box else "")
    sys.stdout.flush()
    return max([abs(rc) for rc in rcs] + [1 if failed else 0])


def build_model(wl, device, prec, batch=None):
    from sgdm_amd.synth import synth_batch, weights_from_seed
    from sgdm_amd.unet import UNetModel, UNetModelCA
    common = dict(image_size=wl["image"], in_channels=3, out_channels=3, model_channels=wl.get("model_channels", 128),
                  num_res_blocks=2,
                  channel_mult=[1, 2, 4], attention_resolutions=[4], num_heads=8, use_scale_shift_norm=True,
                  cond_dim=wl["cond_dim"], condition_method=wl["method"])
    cond = AD(scale_type="imagen")
    if wl["layout_dim"]:
        cond[wl["method"]] = AD(layout_dim=wl["layout_dim"])
    if wl["kind"] == "unet_fast":
        m = UNetModel(dropout=0.1, resblock_updown=True, condition=cond, **common)
    else:
        m = UNetModelCA(dropout=0.0, use_ca_block=True, legacy=False, cond_token_num=1, context_dim=32,
                        use_cls_token_as_pooled=True, condition=cond, **common)
    manifest = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = weights_from_seed(manifest, 23)
    m.load_state_dict(sd)
    m = m.to(device).eval()
    m.hip_precision = prec
    B = batch or wl["batch"]
    data = synth_batch(wl["method"], B, wl["image"], wl["cond_dim"], wl["layout_dim"], seed=23)
    return m, sd, data


def host_cores():
    """threads actually usable: min(affinity, cgroup CPU quota) -- the GPU box shows 256 logical CPUs but
    grants a 16-CPU quota; oversubscribing makes the torch-CPU baseline 10x slower"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(wl, sd, seconds_budget=25.0):
    """the oracle (CPU restatement of the reference UNet, pinned to the reference by tests/golden) timed
    on the host cores: CFG steps at bs=8, extrapolated linearly to the workload's 1000-step images/s."""
    from oracle import unet_ref as U
    from sgdm_amd.synth import synth_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = U.make_cfg(wl["kind"], wl["image"], model_channels=128, cond_dim=wl["cond_dim"],
                     condition_method=wl["method"], layout_dim=wl["layout_dim"],
                     cond_token_num=1 if wl["kind"] == "unetca_fast" else 0,
                     context_dim=32 if wl["kind"] == "unetca_fast" else None)
    B = 8
    data = synth_batch(wl["method"], B, wl["image"], wl["cond_dim"], wl["layout_dim"], seed=23)
    cond = data.get("cond")
    if cond is not None and wl["kind"] == "unetca_fast":
        cond = cond.float()
    x = torch.randn(B, 3, wl["image"], wl["image"], generator=torch.Generator().manual_seed(1))
    t = torch.full((B,), 500, dtype=torch.long)
    times = []
    with torch.no_grad():
        t_start = time.time()
        while len(times) < 3 and (time.time() - t_start) < seconds_budget:
            t0 = time.time()
            U.forward_with_cond_scale(cfg, sd, x, t, 2.0, cond, data.get("layout"))
            times.append(time.time() - t0)
    per_step = min(times)
    out = dict(value=B / (1000.0 * per_step), unit="images/s", cores=cores, kind="port",
               sample=f"{len(times)} CFG UNet steps at bs={B} (UNet batch {2 * B}) of the same model on the host "
                      f"cores, best {per_step:.2f} s/step, extrapolated linearly to 1000 steps",
               s_per_step_bs8=round(per_step, 3))
    # config C1 in full on the CPU path (10-step DDIM, bs=8, ch=64, 32x32)
    from oracle import diffusion_ref as D
    from sgdm_amd.synth import weights_from_seed
    from sgdm_amd.unet import UNetModel
    c1cfg = U.make_cfg("unet_fast", C1["image"], model_channels=C1["model_channels"], cond_dim=C1["cond_dim"],
                       condition_method="label", resblock_updown=True)
    man = [(k, tuple(v)) for k, v, _ in U.param_manifest(c1cfg)]
    sd1 = weights_from_seed(man, 23)
    d1 = synth_batch("label", C1["batch"], C1["image"], C1["cond_dim"], 0, seed=23)
    g = torch.Generator().manual_seed(3)
    xT = torch.randn(C1["batch"], 3, C1["image"], C1["image"], generator=g)
    zs = torch.randn(C1["ddim_steps"], C1["batch"], 3, C1["image"], C1["image"], generator=g)
    with torch.no_grad():
        t0 = time.time()
        D.ddim_sample(D.make_schedule(), lambda x_, t_: U.forward_with_cond_scale(c1cfg, sd1, x_, t_, 2.0, d1["cond"], None),
                      xT, lambda i: zs[i], C1["ddim_steps"], eta=0.0)
        c1_s = time.time() - t0
    out["c1_full"] = dict(seconds=round(c1_s, 2), images=C1["batch"], images_per_s=round(C1["batch"] / c1_s, 3),
                          workload=C1["desc"])
    return out


def pmc_traffic(workload, prec, B):
    """HBM bytes per igemm launch from the PMC passes committed under profiles/ (tools/pmc_hbm.sh: FETCH_SIZE and
    WRITE_SIZE in separate rocprofv3 --pmc runs of this same bench command, gfx950 FETCH_SIZE x2 correction).
    PMC collection needs the profiler around the process, so it cannot be taken live inside this run; null when no
    committed pass matches the workload / precision / batch being benchmarked (file: r<N>_pmc_hbm_<workload>.json at the
    workload's own batch, r<N>_pmc_hbm_<workload>_bs<B>.json otherwise)."""
    if prec != "f16x3":
        return None, None
    stem = workload if B == WORKLOADS[workload]["batch"] else f"{workload}_bs{B}"
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_hbm_{stem}.json")
        if os.path.exists(path):
            break
    else:
        return None, None
    k = json.load(open(path))["kernels"].get("igemm_kernel")
    return (round(k["hbm_bytes_per_launch"]) if k else None,
            f"replayed from {os.path.relpath(path, ROOT)} (separate rocprofv3 --pmc passes of this command: FETCH_SIZE x2 + "
            "WRITE_SIZE; counters cannot be read from inside the process)")


def with_probe(roof, probe, prec):
    """the roofline figure against what this device delivers NOW (bare 16-bit MFMA loop / 3 products; exact mode: the fp32
    MFMA peak has no such probe, the fields stay absent)"""
    if not probe:
        return roof
    roof.update(probe)
    if prec != "f32":
        ceil = probe["device_mfma_tflops"] / 3.0
        roof["device_ceiling_tflops"] = round(ceil, 1)
        roof["frac_of_device_ceiling"] = round(roof["achieved"] / ceil, 4)
        for v in roof["igemm_by_instance"].values():
            v["frac_of_device_ceiling"] = round(v["tflops_per_s"] / ceil, 4)
    return roof


def sampling_roofline(eng, prec, workload, B, probe=None):
    """the `roofline` block of one sampling workload: an instrumented pass over the engine's UNet program (HIP events around
    every launch, on the launch stream), the conv / linear kernel's achieved TFLOP/s against the MFMA peak of the arithmetic
    mode, per template instance, with the committed PMC traffic beside it and -- `probe` = device_probe()'s record -- against
    what this device delivered on a bare MFMA loop in this process"""
    stream = torch.cuda.current_stream().cuda_stream
    agg, inst = {}, {}
    reps = 3
    conv3 = (".in_layers.2", ".out_layers.3", ".op", ".conv", "input_blocks.0.0", "out.2")
    for _ in range(reps):
        for tag, sym, ms, fl, nb in eng.prog.run_profiled(stream):
            a = agg.setdefault(sym, [0.0, 0.0, 0.0, 0])
            a[0] += ms; a[1] += fl; a[2] += nb; a[3] += 1
            if sym == "sgd_igemm":        # the two template instantiations rocprofv3 lists separately
                b = inst.setdefault("taps9_conv3x3" if tag.endswith(conv3) else "taps1_conv1x1_linear", [0.0, 0.0, 0])
                b[0] += ms; b[1] += fl; b[2] += 1
    tot_ms = sum(a[0] for a in agg.values()) / reps
    ig = agg["sgd_igemm"]
    ig_ms, ig_fl, ig_nb, ig_n = ig[0] / reps, ig[1] / reps, ig[2] / reps, ig[3] // reps
    peak = PEAK_TFLOPS[prec]
    ach = ig_fl / (ig_ms * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic(workload, prec, B)
    roof = dict(bound="mfma", kernel="igemm_kernel (fused implicit-GEMM conv/linear, all launches of one UNet eval)",
                achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(ach / peak, 4),
                traffic=traffic, traffic_source=traffic_src, launches_per_step=ig_n,
                launches_per_step_all_kernels=sum(a[3] for a in agg.values()) // reps,
                algorithmic_bytes_per_launch=round(ig_nb / ig_n), avg_launch_ms=round(ig_ms / ig_n, 4),
                igemm_ms_per_step=round(ig_ms, 3), all_kernels_ms_per_step=round(tot_ms, 3),
                algorithmic_gflop_per_step=round(ig_fl / 1e9, 1),
                hbm_algorithmic_frac=round((ig_nb / (ig_ms * 1e-3)) / 8.0e12, 4),
                peak_note=("exact fp32 MFMA peak" if prec == "f32" else
                           "2.5 PF dense 16-bit MFMA / 3 products per fp32-equivalent product"),
                per_kernel_ms={k: round(v[0] / reps, 3) for k, v in sorted(agg.items())},
                igemm_by_instance={k: dict(launches=v[2] // reps, ms_per_step=round(v[0] / reps, 3),
                                           tflops_per_s=round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                                           frac=round(v[1] / (v[0] * 1e-3) / 1e12 / peak, 4))
                                   for k, v in sorted(inst.items())})
    with_probe(roof, probe, prec)
    return roof


def train_step_bench(model, diff, data, cond, layout, B, world, barrier, wl, steps=4, warmup=2):
    """train-step ms (max over ranks): forward_tao -> loss.backward() (bucketed all-reduce inside) -> AdamW -> EMA"""
    import torch.distributed as dist
    from sgdm_amd import ddp
    from sgdm_amd.ema import LitEma
    dev = next(model.parameters()).device
    model.train()
    diff.train()
    from sgdm_amd.optim import FusedAdamWEma
    exchanging = ddp.exchange_active(model)          # world > 1, or a one-rank group with the exchange forced
    if exchanging:
        # replicas start identical (torch DDP's construction-time broadcast; bench seeds make them so anyway): the model
        # BEFORE its EMA shadows are cloned from it
        ddp.sync_initial_state(model)
    ema = LitEma(model)
    # optim/adamw.yaml + data lr/wd; AdamW and the LitEma shadow update run as ONE launch (sgd_adamw_ema_step)
    opt = FusedAdamWEma([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01, ema=ema,
                        ema_model=model)
    x = data["image"].to(dev)

    def step():
        loss, _ = diff.forward_tao(x, cond=cond, layout=layout, cond_drop_prob=0.1)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    # self-verification of the exchange (the first multi-GPU run is the first time this path meets RCCL): after the first
    # step every rank must hold the SAME averaged gradients -- a checksum per rank, gathered
    loss, _ = diff.forward_tao(x, cond=cond, layout=layout, cond_drop_prob=0.1)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    csum = torch.zeros(2, device=dev, dtype=torch.float64)
    for p_ in model.parameters():
        if p_.grad is not None:
            csum[0] += p_.grad.double().sum()
            csum[1] += p_.grad.double().abs().sum()
    opt.step()
    sums = [csum.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(sums, csum)
    for _ in range(max(0, warmup - 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    per_rank = [tt.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, tt)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    model.eval()
    diff.eval()
    ref = sums[0]
    # when did the exchange run relative to the backward program?  (HIP events on the streams the work ran on, last timed
    # step; null when nothing was exchanged)
    overlap = None
    if exchanging:
        for eng in model._engines.values():
            red = getattr(getattr(eng, "backward", None), "reducer", None)
            if eng.n == B and red is not None:
                overlap = red.overlap_stats()
    ov = overlap or {}
    return dict(ms=round(1000.0 * float(tt.item()) / steps, 2), batch_per_gpu=B, global_batch=B * world, steps=steps,
                dropout=float(model.dropout), loss=round(float(loss.item()), 4),
                includes="q_sample + UNet fwd/bwd + " + ("bucketed grad all-reduce on a side stream under the backward + "
                                                          if exchanging else "") + "fused AdamW/EMA step (per-step weight re-pack inside)",
                algorithmic_tflop=round(3 * B * wl["gflop_per_eval_img"] / 1e3, 3),
                world_size=(dist.get_world_size() if dist.is_initialized() else 1),
                backend=(str(dist.get_backend()) if dist.is_initialized() else None),
                exchange=("forced on one rank" if exchanging and world == 1 else bool(exchanging)),
                per_rank_ms=[round(1000.0 * float(t_.item()) / steps, 2) for t_ in per_rank],
                grad_checksum_first_step=[[float(v[0]), float(v[1])] for v in sums],
                # null on one rank: there is nothing to compare, and a trivially true flag reads as a verified exchange
                grad_checksums_equal=(bool(all(torch.equal(v, ref) for v in sums)) if world > 1 else None),
                reserved_cus=int(ddp.reserved_cus(model)),
                exchange_ms=ov.get("exchange_ms"), exposed_exchange_ms=ov.get("exposed_exchange_ms"),
                first_bucket_at_frac_of_backward=ov.get("first_bucket_at_frac_of_backward"),
                exchange_buckets=ov.get("per_bucket"), exchange_backward_ms=ov.get("backward_ms"),
                mbytes_enqueued_after_0p9_of_backward=ov.get("mbytes_enqueued_after_0p9_of_backward"),
                modelled_exposed_ms_at_50GBps=ov.get("modelled_exposed_ms_at_50GBps"),
                modelled_exposed_ms_at_100GBps=ov.get("modelled_exposed_ms_at_100GBps"))


def exchange_probe(args, TB):
    """the data-parallel training step on ONE rank: a world-size-1 "nccl" (= RCCL) group with `hip_force_exchange` on the
    model -- arena, bucket hooks, all-reduce through librccl on the side stream, CU reserve, overlap record.  Runs as a
    CHILD process (RCCL reads its environment once per process and writes a banner to stdout; a hang must not take the
    bench line with it).  Never fatal: what fails is reported as a string."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--exchange-probe-child", "--workload", args.workload, "--prec", args.prec,
           "--train-batch", str(TB)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("EXCHANGE ")]
        if r.returncode != 0 or not lines:
            return dict(error=f"child rc {r.returncode}: {r.stderr[-300:]}")
        return json.loads(lines[-1][len("EXCHANGE "):])
    except Exception as exc:                                 # pragma: no cover - depends on the box
        return dict(error=f"{type(exc).__name__}: {exc}"[:400])


def exchange_probe_child(args):
    import tempfile
    import torch.distributed as dist
    from sgdm_amd.ddp import cap_exchange_channels
    from sgdm_amd.diffusion import LatentDiffusion
    from sgdm_amd.synth import synth_batch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cap_exchange_channels()
    wl = WORKLOADS[args.workload]
    TB, S = args.train_batch, wl["image"]
    with tempfile.TemporaryDirectory() as td:
        dist.init_process_group("nccl", init_method=f"file://{os.path.join(td, 'store')}", rank=0, world_size=1)
        model, _, _ = build_model(wl, dev, args.prec, TB)
        model.hip_force_exchange = True
        diff = LatentDiffusion(device=str(dev), **MODEL_PARAMS)
        diff.set_denoise_fn(model.forward, model.forward_with_cond_scale)
        tdata = synth_batch(wl["method"], TB, S, wl["cond_dim"], wl["layout_dim"], seed=29)
        tcond = tdata.get("cond")
        if tcond is not None:
            tcond = tcond.to(dev) if wl["kind"] == "unet_fast" else tcond.float().to(dev)
        tlayout = tdata["layout"].to(dev) if "layout" in tdata else None

        def barrier():
            torch.cuda.synchronize()
        rec = train_step_bench(model, diff, tdata, tcond, tlayout, TB, 1, barrier, wl, steps=4, warmup=3)
        # The price of the exchange's machinery on ONE device, in ONE process (box-to-box and process-to-process spread is as
        # large as the effect): the same step (a) as above -- arena, buckets, collectives, the CU reserve inside its windows --,
        # (b) with the reserve on every backward launch (round 5), (c) plain single-process (no arena, no reserve);
        # interleaved, three rounds of four steps each.
        eng = [e for e in model._engines.values() if e.n == TB and getattr(e, "backward", None) is not None][0]
        bw_x = eng.backward
        from sgdm_amd.optim import FusedAdamWEma
        opt = FusedAdamWEma([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01)
        x_ = tdata["image"].to(dev)
        model.train(); diff.train()

        def steps_ms(k=4):
            def one():
                loss, _ = diff.forward_tao(x_, cond=tcond, layout=tlayout, cond_drop_prob=0.1)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
            one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                one()
            torch.cuda.synchronize()
            return 1000.0 * (time.perf_counter() - t0) / k
        res = dict(windows=[], every_launch=[], plain=[])
        bw_plain = None
        for _ in range(3):
            model.hip_ddp, eng.backward = True, bw_x
            os.environ["SGDM_RESERVE_WINDOWS"] = "1"; bw_x.apply_grid_cap()
            res["windows"].append(steps_ms())
            os.environ["SGDM_RESERVE_WINDOWS"] = "0"; bw_x.apply_grid_cap()
            res["every_launch"].append(steps_ms())
            model.hip_ddp, eng.backward = False, bw_plain
            res["plain"].append(steps_ms())
            bw_plain = eng.backward
        os.environ["SGDM_RESERVE_WINDOWS"] = "1"
        model.hip_ddp, eng.backward = True, bw_x
        eng._grid_cap = -1                                # (force the next set_grid_cap to re-apply)
        eng.set_grid_cap(16)
        rec["same_process_ms"] = {k: round(sorted(v)[1], 2) for k, v in res.items()}
        caps = [int(a_.grid_cap) for a_, _ in bw_x.late]
        rec["reserve_windows"] = dict(capped_launches=sum(1 for c_ in caps if c_ > 0), conv_launches=len(caps),
                                      program_entries_inside=list(bw_x._windows or ()))
        dist.destroy_process_group()
    keep = ("ms", "batch_per_gpu", "steps", "world_size", "backend", "exchange", "reserved_cus", "exchange_ms",
            "exposed_exchange_ms", "first_bucket_at_frac_of_backward", "exchange_backward_ms", "exchange_buckets",
            "mbytes_enqueued_after_0p9_of_backward", "modelled_exposed_ms_at_50GBps", "modelled_exposed_ms_at_100GBps",
            "same_process_ms", "reserve_windows")
    print("EXCHANGE " + json.dumps({k: rec.get(k) for k in keep}), flush=True)


def device_probe(dev, seconds=0.15):
    """what THIS device delivers right now: the boxes of the pool differ by 5-7 % for one binary
    and the chip trades clock for matrix-pipe duty, so the roofline fraction is reported a second time against a ceiling
    measured in this process, seconds after the timed steps: a bare v_mfma_f32_16x16x32_f16 loop on random register operands
    (one wave per SIMD on every CU, csrc/tools/probe.hip) and 16-byte-per-lane copy / read / write streams over 256 MiB."""
    from sgdm_amd import _lib as L
    lib = L.load_tools()                                  # diagnostics library (include/sgdm_hip_tools.h), not the product's
    st = torch.cuda.current_stream().cuda_stream
    cus = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    sink = torch.empty(cus * 256, device=dev)

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3
    iters = 20000
    run = lambda: L.check(lib.sgd_debug_mfma_probe(cus, iters, 12345, 0, C_void(sink), st), "mfma_probe")
    run()
    t = timed(run)
    iters = max(1000, int(iters * seconds / max(t, 1e-6)))          # one launch of ~`seconds`: the clock settles under the load
    t = timed(run)
    mfma_tf = float(lib.sgd_debug_mfma_probe_flops(cus, iters, 0)) / t / 1e12
    # HBM twin of the MFMA probe.  The round-5 copy probe read 4.6-4.8 TB/s against the guide's 6.29
    # because of its ACCESS SHAPE, not its size: a lane's four loads were 8 MB apart (grid-stride).  tools/hbm_probe_sweep.py
    # (profiles/r6_hbm_probe_sweep.txt): every wave walking contiguous 8 KiB pieces with non-temporal accesses copies 256 MiB
    # at 6.2-6.5 TB/s, reads at 6.6 and writes at 6.9 -- the figures reported here (variant 4 / 5 / 6 of the probe, grid 8192).
    n = 64 << 20                                                      # floats: 256 MiB read + 256 MiB written per pass
    src, dst = torch.randn(n, device=dev), torch.empty(n, device=dev)

    def rate(variant, grid, nbytes):
        fn = lambda: [L.check(lib.sgd_debug_copy_probe(C_void(src), C_void(dst), n, variant, grid, st), "copy_probe") for _ in range(10)]
        fn()
        return 10 * nbytes / timed(fn) / 1e12
    copy_tbps = rate(4, 8192, 2 * 4.0 * n)
    assert torch.equal(src[-4096:], dst[-4096:])
    stride_tbps = rate(0, 2048, 2 * 4.0 * n)
    read_tbps = rate(5, 2048, 4.0 * n)
    write_tbps = rate(6, 8192, 4.0 * n)
    return dict(device_mfma_tflops=round(mfma_tf, 1), device_mfma_probe_ms=round(t * 1e3, 1),
                device_copy_tbps=round(copy_tbps, 3), device_read_tbps=round(read_tbps, 3), device_write_tbps=round(write_tbps, 3),
                device_copy_tbps_grid_stride=round(stride_tbps, 3), device_cus=cus)


def C_void(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def train_step_roofline(model, prec, B, S):
    """per-kernel record of ONE training step's device programs: the forward program in
    train mode and the backward program replayed with a HIP event pair around every launch (events on the launch
    stream), on the activations the last timed step left behind.  Single rank only (the backward program's bucket hooks
    would start collectives).  MFMA kernels carry TF/s and the fraction of the arithmetic mode's peak."""
    from sgdm_amd import _lib as L
    eng = model._engines.get((B, S, S, L.PREC_BY_NAME[prec]))
    if eng is None or getattr(eng, "backward", None) is None:
        return None
    stream = torch.cuda.current_stream().cuda_stream
    peak = PEAK_TFLOPS[prec]
    rec, tot = {}, 0.0
    for phase, prog in (("fwd", eng.prog), ("bwd", eng.backward.prog)):
        for tag, sym, ms, fl, nb in prog.run_profiled(stream):
            if sym == "bucket_ready":
                continue
            key = sym if phase == "fwd" or sym != "sgd_igemm" else "sgd_igemm(dgrad)"
            if sym == "sgd_igemm" and phase == "fwd":
                key = "sgd_igemm(forward)"
            r = rec.setdefault(key, [0.0, 0.0, 0])
            r[0] += ms; r[1] += fl; r[2] += 1
            tot += ms
    out = {}
    for k, (ms, fl, n) in sorted(rec.items(), key=lambda kv: -kv[1][0]):
        e = dict(ms=round(ms, 3), launches=n)
        if fl > 0:
            tf = fl / (ms * 1e-3) / 1e12
            e.update(tflops_per_s=round(tf, 1), frac=round(tf / peak, 4))
        out[k] = e
    mfma_ms = sum(v[0] for k, v in rec.items() if v[1] > 0)
    mfma_fl = sum(v[1] for k, v in rec.items() if v[1] > 0)
    return dict(bound="mfma", peak=round(peak, 1), unit="TFLOP/s", device_ms_per_step=round(tot, 3),
                mfma_kernels_ms=round(mfma_ms, 3), achieved=round(mfma_fl / (mfma_ms * 1e-3) / 1e12, 1),
                frac=round(mfma_fl / (mfma_ms * 1e-3) / 1e12 / peak, 4), per_kernel=out,
                note="forward (train mode) + backward programs with HIP events around every launch; excludes the "
                     "per-step weight re-pack, the optimizer and host gaps that the wall-clock `ms` includes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prec", default=os.environ.get("SGDM_PREC", "f16x3"))
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--train-batch", type=int, default=80, help="per-GPU batch of the train-step leg (metric: bs=80)")
    ap.add_argument("--no-train40", action="store_true", help="skip the bs=40/GPU train-step leg (configs[2])")
    ap.add_argument("--no-exchange-probe", action="store_true",
                    help="skip the one-rank RCCL run of the data-parallel step (train_step.exchange_world1)")
    ap.add_argument("--no-extra", action="store_true", help="skip the f32_exact / c5 / c1 sub-records")
    ap.add_argument("--no-full", action="store_true", help="skip the complete 1000-step trajectory (full_trajectory)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of the hipGraph-captured step")
    ap.add_argument("--exchange-probe-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.exchange_probe_child:
        return exchange_probe_child(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    import torch.distributed as dist
    # bind the rank to its GPU BEFORE the process group exists: RCCL picks up the current device at its first collective
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  SGDM_DIST_BACKEND=gloo lets several ranks share ONE GPU to exercise this path on a
        # single-GPU box (tests only: the ranks then time-slice the device)
        backend = os.environ.get("SGDM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            # the exchange's kernels get the CUs the training programs leave free (sgdm_amd.ddp.reserved_cus) and no more:
            # the PRODUCT's own rule (the same call HipDDPStrategy.setup_environment makes), before the group exists
            from sgdm_amd.ddp import cap_exchange_channels
            cap_exchange_channels()
        dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)

    from sgdm_amd.diffusion import LatentDiffusion
    wl = WORKLOADS[args.workload]
    B = args.batch or wl["batch"]
    model, sd, data = build_model(wl, dev, args.prec, B)
    diff = LatentDiffusion(device=str(dev), **MODEL_PARAMS)
    diff.set_denoise_fn(model.forward, model.forward_with_cond_scale)
    S = wl["image"]
    cond = data.get("cond")
    if cond is not None:
        cond = cond.to(dev) if wl["kind"] == "unet_fast" else cond.float().to(dev)
    layout = data["layout"].to(dev) if "layout" in data else None
    dkw = dict(cond=cond, layout=layout, cond_scale=2.0)
    skw = dict(sampling_method="native", num_timesteps=1000, ddim_eta=0.0, log_num_per_prog=10, clip_denoised=True,
               dtp=1, temperature=1.0, noise_dropout=0, random_sample_condition=False, return_inter_dict=True,
               hip_graph=not args.no_graph)
    torch.manual_seed(23 + rank)
    x = torch.randn(B, 3, S, S, device=dev)

    def run_steps(x, idx):
        img, _ = diff.sampler.sample((B, 3, S, S), sampling_kwargs=skw, denoise_sample_fn=diff.denoise_sample_fn,
                                     denoise_sample_fn_kwargs=dkw, x_T=x, step_indices=idx)
        return img

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # one-time setup outside the timed region even with --warmup 0: building the static launch program, allocating its
        # workspace and packing the weights happen on the first evaluation (result discarded)
        run_steps(x.clone(), [999])
        # step i of the run is timestep 999 - i (wrapping past 0 so any --steps / --warmup is valid)
        x = run_steps(x, [(999 - i) % 1000 for i in range(args.warmup)])     # W untimed warm-up steps
        barrier()
        t0 = time.perf_counter()
        x = run_steps(x, [(999 - i) % 1000 for i in range(args.warmup, args.warmup + args.steps)])   # EXACTLY K steps
        barrier()
        elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    ms_per_step = 1000.0 * elapsed / args.steps
    value = world * B / (1000.0 * (elapsed / args.steps))
    assert torch.isfinite(x).all()

    # ---- the metric in full: ONE complete 1000-step trajectory through the reference's entry
    # point -- LatentDiffusion.p_sample_loop('native') (ddpm.py:108-122 -> ddpm_sampler.py:194-238): 1000 captured steps,
    # the 9 pred_x0 / x_inter snapshots and the uint8 tail -- wall clock, max over ranks.  `value` above stays the K timed
    # steps the bench contract prescribes; this is the steady-state figure beside it (~18 s per GPU).
    full = None
    if not args.no_full:
        with torch.no_grad():
            barrier()
            t0 = time.perf_counter()
            u8, inter = diff.p_sample_loop("native", (B, 3, S, S), skw, denoise_sample_fn_kwargs=dict(dkw), condition_kwargs={})
            barrier()
            ft = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(ft, op=dist.ReduceOp.MAX)
        fsec = float(ft.item())
        assert u8.dtype == torch.uint8 and tuple(inter["pred_x0"].shape) == (9, B, 3, S, S)
        full = dict(seconds=round(fsec, 3), images=world * B, value=round(world * B / fsec, 4), unit="images/s",
                    ms_per_step=round(fsec, 4), steps=1000,
                    vs_k_step_extrapolation=round((world * B / fsec) / value, 4),
                    includes="1000 hipGraph-replayed CFG steps + per-step RNG draw + 9 snapshot pairs + uint8 conversion "
                             "(LatentDiffusion.p_sample_loop('native'))")

    # (right after the timed region, before the training / extra legs heat the device: the instrumented pass sees the
    # clocks the timed steps saw)
    roof, probe = None, None
    PREC_ID = __import__("sgdm_amd._lib", fromlist=["x"]).PREC_BY_NAME[args.prec]
    if rank == 0 and not args.no_profile:
        # ---- instrumented pass: HIP events around every launch of the UNet program (same stream), and the device's own
        # ceiling measured seconds later in this process
        # (the pass first: the probe is a second of full-power MFMA, and what follows it reads a hotter device)
        roof = sampling_roofline(model._engines[(2 * B, S, S, PREC_ID)], args.prec, args.workload, B)
        try:
            probe = device_probe(dev)
            with_probe(roof, probe, args.prec)
        except Exception as exc:                          # a diagnostic must not cost the bench line
            roof["device_probe_error"] = f"{type(exc).__name__}: {exc}"[:200]

    # ---- second half of BASELINE.json's metric: DDPM train-step time (q_sample + UNet fwd/bwd + RCCL gradient
    # all-reduce overlapped with backward + AdamW + EMA), per-GPU batch of the config, dropout as configured
    train = train40 = None
    if not args.no_train:
        from sgdm_amd.synth import synth_batch
        TB = args.train_batch
        tdata = synth_batch(wl["method"], TB, S, wl["cond_dim"], wl["layout_dim"], seed=29 + rank)
        tcond = tdata.get("cond")
        if tcond is not None:
            tcond = tcond.to(dev) if wl["kind"] == "unet_fast" else tcond.float().to(dev)
        tlayout = tdata["layout"].to(dev) if "layout" in tdata else None
        train = train_step_bench(model, diff, tdata, tcond, tlayout, TB, world, barrier, wl)
        if world == 1 and not args.no_profile:
            train["roofline"] = train_step_roofline(model, args.prec, TB, S)
        # ---- BASELINE.json configs[2] (C3): the same model at bs=40 per GPU
        if TB != 40 and not args.no_train40:
            sl = lambda v: v[:40] if v is not None else None
            t40 = {k: sl(v) for k, v in tdata.items()}
            train40 = train_step_bench(model, diff, t40, sl(tcond), sl(tlayout), 40, world, barrier, wl)
        # ---- one rank: the data-parallel step's own code on the hardware at hand -- a world-size-1 RCCL group with the
        # exchange forced (arena, bucketed all-reduce through librccl on the side stream, CU reserve): what the N > 1 legs
        # run, with its overlap record.  A record of the path, not a scaling number.
        # (never from a traced process: the child would be started from a GPU-initialised parent with the profiler's
        # preload in its environment -- ADVICE round 5)
        traced = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES"))
        if world == 1 and not args.no_exchange_probe and not traced:
            train["exchange_world1"] = exchange_probe(args, TB)

    def time_sampling(mdl, dif, bsz, size, kw, steps, warm, skw_):
        """ms per CFG sampling step of `mdl` (setup outside, W untimed + K timed steps, barrier + sync around)"""
        xx = torch.randn(bsz, 3, size, size, device=dev)

        def go(x_, idx):
            return dif.sampler.sample((bsz, 3, size, size), sampling_kwargs=skw_, denoise_sample_fn=dif.denoise_sample_fn,
                                      denoise_sample_fn_kwargs=kw, x_T=x_, step_indices=idx)[0]
        with torch.no_grad():
            go(xx.clone(), [999])
            xx = go(xx, [(999 - i) % 1000 for i in range(warm)])
            barrier()
            t0_ = time.perf_counter()
            go(xx, [(999 - i) % 1000 for i in range(warm, warm + steps)])
            barrier()
            return 1000.0 * (time.perf_counter() - t0_) / steps

    extra = {}
    if not args.no_extra and args.workload == "c2" and world == 1:
        # ---- (first of the extras: measured after the bs=80 models below it read 80 ms instead of 29 -- host-side allocator
        # churn of the models just freed, not device time)
        # ---- BASELINE.json configs[0] (C1) at its true shape on the GPU: 10-step DDIM, bs=8 -- ~140 small launches per
        # step, so the host launch path matters: eager ctypes launches vs the hipGraph-captured step
        m1, _, d1 = build_model(C1, dev, args.prec, C1["batch"])
        diff1 = LatentDiffusion(device=str(dev), **MODEL_PARAMS)
        diff1.set_denoise_fn(m1.forward, m1.forward_with_cond_scale)
        k1 = dict(cond=d1["cond"].to(dev), layout=None, cond_scale=2.0)
        sk1 = dict(skw, sampling_method="ddim", num_timesteps=C1["ddim_steps"])
        c1 = {}
        with torch.no_grad():
            for name, g_on in (("eager", False), ("graph", True)):
                kw_ = dict(sk1, hip_graph=g_on)
                shape1 = (C1["batch"], 3, C1["image"], C1["image"])
                run1 = lambda: diff1.p_sample_loop("ddim", shape1, kw_, denoise_sample_fn_kwargs=dict(k1), condition_kwargs={})
                run1()
                ts_ = []
                for _ in range(9):                      # median of 9 whole trajectories (host hiccups are outliers here)
                    barrier()
                    t0_ = time.perf_counter()
                    run1()
                    barrier()
                    ts_.append(1000.0 * (time.perf_counter() - t0_))
                c1[name + "_ms_per_trajectory"] = round(sorted(ts_)[len(ts_) // 2], 3)
                c1[name + "_ms_min"] = round(min(ts_), 3)
        c1.update(workload=C1["desc"], images_per_s=round(C1["batch"] / (c1["graph_ms_per_trajectory"] * 1e-3), 1),
                  graph_speedup=round(c1["eager_ms_per_trajectory"] / c1["graph_ms_per_trajectory"], 2),
                  note="whole p_sample_loop (10 steps, all of them snapshot steps, uint8 conversion); the captured step is cached on the model, capture cost excluded by the warm-up call")
        extra["c1"] = c1
        del m1, diff1
        # ---- the headline workload in exact-fp32 MFMA arithmetic (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 FMA chain)
        if args.prec != "f32":
            model.hip_precision = "f32"
            ms32 = time_sampling(model, diff, B, S, dkw, 3, 1, skw)
            tfl = 2 * B * wl["gflop_per_eval_img"] / ms32
            extra["f32_exact"] = dict(ms_per_step=round(ms32, 3), value=round(B / ms32, 4), unit="images/s",
                                      tflops_per_s=round(tfl, 1), frac_of_fp32_mfma_peak=round(tfl / PEAK_TFLOPS["f32"], 4),
                                      peak=PEAK_TFLOPS["f32"], steps=3, warmup=1)
            model.hip_precision = args.prec
        # ---- BASELINE.json configs[4] (C5): unetca_fast stegoclusterlayout bs=80 -- sampling step + train step
        w5 = WORKLOADS["c5"]
        m5, _, d5 = build_model(w5, dev, args.prec, w5["batch"])
        diff5 = LatentDiffusion(device=str(dev), **MODEL_PARAMS)
        diff5.set_denoise_fn(m5.forward, m5.forward_with_cond_scale)
        k5 = dict(cond=d5["cond"].float().to(dev), layout=d5["layout"].to(dev), cond_scale=2.0)
        ms5 = time_sampling(m5, diff5, w5["batch"], S, k5, 5, 2, skw)
        extra["c5"] = dict(workload=w5["desc"], ms_per_step=round(ms5, 3), value=round(w5["batch"] / ms5, 4), unit="images/s",
                           tflops_per_s=round(2 * w5["batch"] * w5["gflop_per_eval_img"] / ms5, 1), steps=5, warmup=2)
        if not args.no_profile:      # (the roofline at the batch north_star names, per workload)
            extra["c5"]["roofline"] = sampling_roofline(m5._engines[(2 * w5["batch"], S, S, PREC_ID)], args.prec, "c5", w5["batch"], probe)
        if not args.no_train:
            extra["c5"]["train_step"] = train_step_bench(m5, diff5, d5, k5["cond"], k5["layout"], w5["batch"], world, barrier, w5)
        del m5, diff5
        gc.collect(); torch.cuda.empty_cache()
        # ---- BASELINE.json configs[3] (C4): VOC-64 unetca_fast self-boxed clusterlayout (LOST), cond_dim 100, bs=80
        w4 = WORKLOADS["c4"]
        m4, _, d4 = build_model(w4, dev, args.prec, w4["batch"])
        diff4 = LatentDiffusion(device=str(dev), **MODEL_PARAMS)
        diff4.set_denoise_fn(m4.forward, m4.forward_with_cond_scale)
        k4 = dict(cond=d4["cond"].float().to(dev), layout=d4["layout"].to(dev), cond_scale=2.0)
        ms4 = time_sampling(m4, diff4, w4["batch"], S, k4, 5, 2, skw)
        extra["c4"] = dict(workload=w4["desc"], ms_per_step=round(ms4, 3), value=round(w4["batch"] / ms4, 4), unit="images/s",
                           tflops_per_s=round(2 * w4["batch"] * w4["gflop_per_eval_img"] / ms4, 1), steps=5, warmup=2)
        if not args.no_profile:
            extra["c4"]["roofline"] = sampling_roofline(m4._engines[(2 * w4["batch"], S, S, PREC_ID)], args.prec, "c4", w4["batch"], probe)
        if not args.no_train:
            extra["c4"]["train_step"] = train_step_bench(m4, diff4, d4, k4["cond"], k4["layout"], w4["batch"], world, barrier, w4)
        del m4, diff4
        gc.collect(); torch.cuda.empty_cache()
        # ---- the headline workload at C5's batch (bs=80, UNet batch 160): C2 and C5 side by side at one batch size
        if B != w5["batch"]:
            m2, _, d2 = build_model(wl, dev, args.prec, w5["batch"])
            diff2 = LatentDiffusion(device=str(dev), **MODEL_PARAMS)
            diff2.set_denoise_fn(m2.forward, m2.forward_with_cond_scale)
            k2 = dict(cond=d2["cond"].to(dev), layout=None, cond_scale=2.0)
            ms2 = time_sampling(m2, diff2, w5["batch"], S, k2, 5, 2, skw)
            extra["c2_bs80"] = dict(workload=wl["desc"].replace("bs=40/GPU (UNet batch 80)", "bs=80/GPU (UNet batch 160)"),
                                    ms_per_step=round(ms2, 3), value=round(w5["batch"] / ms2, 4), unit="images/s",
                                    tflops_per_s=round(2 * w5["batch"] * wl["gflop_per_eval_img"] / ms2, 1), steps=5, warmup=2)
            if not args.no_profile:
                extra["c2_bs80"]["roofline"] = sampling_roofline(m2._engines[(2 * w5["batch"], S, S, PREC_ID)], args.prec,
                                                                 args.workload, w5["batch"], probe)
            del m2, diff2
            gc.collect(); torch.cuda.empty_cache()

    out = None
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(wl, sd)
        out = {
            "metric": "cfg_sampled_images_per_sec_64x64_1000step_ddpm", "value": round(value, 4), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f16x3": "f32 as 3xf16 split MFMA products, f32 accumulate",
                      "bf16x3": "f32 as 3xbf16 split MFMA products, f32 accumulate"}[args.prec],
            "data": "synthetic",
            "config": {"workload": wl["desc"], "batch_per_gpu": B, "unet_batch": 2 * B, "precision_mode": args.prec,
                       "algorithmic_tflop_per_step": round(2 * B * wl["gflop_per_eval_img"] / 1e3, 3),
                       "launch": "eager" if args.no_graph else "hipGraph-captured step"},
            "roofline": roof, "cpu_baseline": cpu, "train_step": train, "train_step_bs40": train40,
            "full_trajectory": full,
        }
        out.update(extra)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # ONE line, and the last thing on stdout: whatever native libraries hold in C stdio buffers (RCCL writes a version
        # banner to stdout, flushed at exit -- i.e. AFTER a line printed here) goes out first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
