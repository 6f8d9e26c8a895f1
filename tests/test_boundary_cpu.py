"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares, the drop-in modules resolve through the reference's `target:` strings with reference-identical
state_dict manifests, the plugin surface / EMA / LR schedule / schedule tables match the golden vectors,
and the product path refuses to run without the GPU (no CPU fallback, no oracle import)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT, load_json, load_npz

INDEX = load_json("unet_index.json")


class AD(dict):
    __getattr__ = dict.__getitem__


def test_library_exports_every_declared_symbol():
    from sgdm_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "sgdm_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(sgd_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = L.load()                                    # binds every symbol or raises
    assert lib.sgd_abi_version() == L.ABI_VERSION
    assert ctypes.sizeof(L.IgemmArgs) == 216
    # the library is a build of THIS tree: the source digest it embeds equals the digest of the sources on disk
    import importlib.util
    spec = importlib.util.spec_from_file_location("sgdm_build", os.path.join(ROOT, "self-guided-diffusion-models_amd", "build.py"))
    bld = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bld)
    assert lib.sgd_build_id().decode() == bld.source_id()
    # argument validation needs no GPU: invalid descriptors are rejected before any launch
    assert lib.sgd_igemm(None, None) == 1
    assert lib.sgd_packed_weight_bytes(128, 128, 3, 0) == 9 * 128 * 128 * 4
    assert lib.sgd_packed_weight_bytes(3, 30, 3, 1) == 9 * 32 * 32 * 4


def test_diagnostics_live_in_their_own_library():
    """sgd_debug_* (device calibration, CU stand-in, stream probes) are not part of the release
    library; libsgdm_hip_tools.so exports exactly what include/sgdm_hip_tools.h declares and the product never loads it"""
    from sgdm_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "sgdm_hip_tools.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t)\s+(sgd_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(L.TOOLS_SIGNATURES), declared ^ set(L.TOOLS_SIGNATURES)
    L.load_tools()                                    # binds every symbol or raises
    assert b"sgd_debug_" not in open(L.LIB_PATH, "rb").read()
    assert "sgd_debug" not in open(os.path.join(ROOT, "include", "sgdm_hip.h")).read()
    pkg = os.path.join(ROOT, "self-guided-diffusion-models_amd")
    for base, _, files in os.walk(pkg):
        if "csrc" in base or "__pycache__" in base:
            continue
        for f in files:
            if f.endswith(".py") and f != "_lib.py":
                assert "load_tools" not in open(os.path.join(base, f)).read(), f


def test_host_side_launch_rules_need_no_gpu():
    """round 6: rules that live on the host side of the C-ABI answer without a device -- the column tile of small launches (through
    the statistics-slot count it implies), the shapes the row-stream GroupNorm-backward reduce serves, and the argument checks of the
    exchange entry (no RCCL library is touched for an invalid call)"""
    from sgdm_amd import _lib as L
    lib = L.load()
    a = L.IgemmArgs()
    a.mode, a.c0, a.stride, a.prec = L.MODE_CONV3, 128, 1, L.PREC_F16X3

    def parts(n, hw, cout, tune=0):
        a.n, a.hi, a.wi, a.ho, a.wo, a.cout, a.y_ld, a.tune = n, hw, hw, hw, hw, cout, cout, tune
        return lib.sgd_igemm_stats_parts(ctypes.byref(a))
    # 128-row tiles of one image: hw^2 / 128 slots per wave row; one wave row on the 128-column tile, four on the 32-column instance
    assert parts(80, 64, 128) == 32 and parts(80, 16, 512) == 2          # the benchmark batch: 128-column tiles
    assert parts(2, 16, 128) == 8 and parts(2, 16, 128, L.TUNE_NO_SMALL) == 2     # 4 tiles of 128 columns: the 32-column instance
    assert parts(16, 16, 256) == 8 and parts(32, 16, 256) == 2          # 64 such tiles: still small; 128: not any more
    assert parts(80, 8, 256) == 0                                       # two images per tile: no fused statistics
    assert parts(80, 64, 96) == 128                                     # not a multiple of 128 channels: always the 32-column instance
    f = lib.sgd_gn_bwd_rows_chunks
    assert f(80, 64, 64, 128) == 16 and f(80, 32, 32, 256) == 16 and f(80, 32, 32, 32) == 4 and f(80, 16, 16, 512) == 0
    assert f(80, 64, 64, 96) == 0 and f(80, 64, 64, 2048) == 0
    assert lib.sgd_allreduce_bucket(None, None, 0, None) == 1
    assert lib.sgd_adamw_ema_step(None, None, 0, 0, 1e-3, 0.1, 0.999, 1e-3, 1e-8, 0.0, 1.0, 1.0, -1.0, None, None) == 1


def test_release_library_reads_no_environment():
    """"no global state, re-entrant per stream" (VERDICT round 4, weak #7).  The shipped library neither
    imports getenv nor carries the name of a tuning variable: schedule overrides are fields of the call
    (sgd_igemm_args.tune / grid_cap), experiment code does not live in the tree (git history: 4ce7de0 and before)."""
    from sgdm_amd import _lib as L
    data = open(L.LIB_PATH, "rb").read()
    assert b"getenv" not in data
    assert b"SGDM_" not in data
    for src in ("igemm.hip", "igemm_host.hip", "igemm_shared.h", "pack.hip", "exchange.hip", "backward.hip", "attention.hip", "misc.hip",
                "norm.hip", "narrow.hip"):
        text = open(os.path.join(ROOT, "self-guided-diffusion-models_amd", "csrc", src)).read()
        outside_probe = re.sub(r"#ifdef SGDM_PROBE\n.*?#endif", "", text, flags=re.S)
        assert "getenv(" not in outside_probe, src
    assert L.IgemmArgs.tune.offset == L.IgemmArgs.grid_cap.offset + 4


def _kernel_resources(obj):
    """(name, vgpr spills, scratch bytes) of every gfx950 kernel in one object of csrc/build (tools/kernel_resources.sh)"""
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    path = os.path.join(ROOT, "self-guided-diffusion-models_amd", "csrc", "build", obj)
    if not (os.path.exists(path) and os.path.exists(os.path.join(llvm, "llvm-readelf"))):
        pytest.skip("object file or llvm tools missing")
    with tempfile.TemporaryDirectory() as t:
        subprocess.check_call([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={t}/fat.bin", path])
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={t}/fat.bin",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={t}/dev.co"])
        notes = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", f"{t}/dev.co"], text=True)
    out = []
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        out.append((g("name"), int(g("vgpr_spill_count")), int(g("private_segment_fixed_size"))))
    return out


def test_hot_kernels_have_no_scratch():
    """A kernel with any scratch pays ~3 us on every launch, and a spill reload inside a pipelined loop is a vector memory
    operation the in-order vmcnt counts behind the loads it was meant to overlap (round 5: the 1x1 weight-gradient kernel lost
    up to a third of its rate that way).  The instances the sampler and the training step launch per layer must stay clean:
    every instance of the conv kernel, the wave-specialised 3x3 weight gradient, the pipelined 1x1 weight-gradient forms."""
    for obj in ("igemm_f16x3.o", "igemm_bf16x3.o", "igemm_f32.o"):
        ks = [k for k in _kernel_resources(obj) if "igemm_kernel" in k[0]]
        assert ks, obj
        assert all(sp == 0 and scr == 0 for _, sp, scr in ks), [k for k in ks if k[1] or k[2]]
    ks = _kernel_resources("backward.o")
    ws = [k for k in ks if "wgrad_conv_ws_kernel" in k[0]]
    flat = [k for k in ks if re.search(r"wgrad_conv_kernelILi\dELb1ELi1ELi[123]E", k[0])]
    assert len(ws) == 4 and len(flat) == 6, (len(ws), len(flat))
    assert all(sp == 0 and scr == 0 for _, sp, scr in ws + flat), [k for k in ws + flat if k[1] or k[2]]


def _build(name):
    entry = INDEX[name]
    kw = dict(entry["ctor"])
    cond = AD(scale_type="imagen")
    if entry["layout_dim"]:
        cond[kw["condition_method"]] = AD(layout_dim=entry["layout_dim"])
    target = ("dynamic.diffusionmodules.openaimodel.UNetModel" if entry["kind"] == "unet_fast"
              else "dynamic.diffusionmodules.openaimodel_ca.UNetModel")      # config/dynamic/*.yaml:1
    from sgdm_amd.util import instantiate_from_config
    return instantiate_from_config(dict(target=target, params=dict(condition=cond, **kw))), entry


@pytest.fixture(scope="module", autouse=True)
def dropin_path():
    p = os.path.join(PKG, "dropin")
    sys.path.insert(0, p)
    for m in [k for k in sys.modules if k.split(".")[0] in ("dynamic", "diffusion", "dynamic_input", "diffusion_utils")]:
        del sys.modules[m]
    yield
    sys.path.remove(p)


@pytest.mark.parametrize("name", sorted(INDEX))
def test_dropin_targets_and_manifest(name):
    m, entry = _build(name)
    sd = m.state_dict()
    params = dict(m.named_parameters())
    mine = [[k, list(v.shape), ("param" if params[k].requires_grad else "frozen") if k in params else "buffer"]
            for k, v in sd.items()]
    assert mine == entry["manifest"]
    # zero-initialised tensors of the reference (openaimodel.py:273-276,357,833-834)
    assert float(sd["out.2.weight"].abs().sum()) == 0.0
    assert float(sd["middle_block.0.out_layers.3.weight"].abs().sum()) == 0.0


def test_no_cpu_fallback():
    m, entry = _build("uf_label_c32_s16")
    x = torch.zeros(2, 3, 16, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, torch.zeros(2, dtype=torch.long), cond=torch.zeros(2, 10), cond_drop_prob=0.0)


def test_product_never_imports_oracle():
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), os.path.join(root, f)


def test_plugin_surface_matches_reference():
    from dynamic_input.condition import prepare_condition_kwargs, prepare_denoise_fn_kwargs_4sampling
    gold = load_json("condition_plugin.json")
    names = ("label", "cluster", "lostbboxmask", "segmask", "stegomask", "stego_attr", "id")

    def mk(method, how, training):
        hp = AD(cond_dim=5, condition_method=method, cond_drop_prob=0.1,
                condition=AD(clusterlayout=AD(how=how), layout=AD(how=how)))
        return AD(hparams=hp, training=training, device="cpu")

    for key, want in gold.items():
        parts = key.split("|")
        b = {k: torch.ones(2, 3) * (i + 1) for i, k in enumerate(names)}
        if parts[0] == "sampling":
            method = None if parts[1] == "None" else parts[1]
            how = None if parts[2] == "None" else parts[2]
            r = prepare_denoise_fn_kwargs_4sampling(mk(method, how, False), b, dict(random_sample_condition=False), 2.0)
            assert sorted(r.keys()) == want
            continue
        method = None if parts[0] == "None" else parts[0]
        how = None if parts[1] == "None" else parts[1]
        r = prepare_condition_kwargs(mk(method, how, bool(int(parts[2]))), b)
        got = {}
        for k, v in r.items():
            got[k] = [n for n, bv in b.items() if torch.equal(bv.float(), v.float())][0] if torch.is_tensor(v) else v
        assert got == want, key
    with pytest.raises(ValueError):
        prepare_condition_kwargs(mk("nonsense", None, True), {})


def test_litema_matches_reference():
    from dynamic.ema import LitEma
    v = load_npz("diffusion.npz")
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    shapes = [tuple(p.shape) for p in lin.parameters()]
    sizes = [int(np.prod(s)) for s in shapes]
    with torch.no_grad():
        for p, a in zip(lin.parameters(), np.split(v["ema.init"], np.cumsum(sizes)[:-1])):
            p.copy_(torch.from_numpy(a.copy()).reshape(p.shape))
    ema = LitEma(lin)
    deltas = np.split(v["ema.deltas"], np.cumsum(sizes * 3)[:-1])
    for step in range(3):
        with torch.no_grad():
            for i, p in enumerate(lin.parameters()):
                p.add_(torch.from_numpy(deltas[step * 4 + i].copy()).reshape(p.shape))
        ema(lin)
    assert list(ema.m_name2s_name.values()) == list(v["ema.keys"])
    got = np.concatenate([dict(ema.named_buffers())[s].numpy().ravel() for s in ema.m_name2s_name.values()])
    assert int(ema.num_updates) == int(v["ema.num_updates"])
    assert np.allclose(got, v["ema.shadow"], rtol=0, atol=1e-6)
    # store / copy_to / restore round trip
    ema.store(lin.parameters())
    ema.copy_to(lin)
    assert np.allclose(np.concatenate([p.detach().numpy().ravel() for p in lin.parameters()]), got)
    ema.restore(lin.parameters())


def test_lr_lambda_matches_reference():
    from sgdm_amd.util import LambdaLinearScheduler
    v = load_npz("diffusion.npz")
    sch = LambdaLinearScheduler(warm_up_steps=[500], cycle_lengths=[10000000000000], f_start=[1.e-6], f_max=[1.],
                                f_min=[1.])
    got = np.asarray([sch.schedule(int(n)) for n in v["lr.n"]], dtype=np.float64)
    assert np.array_equal(got, v["lr.f"])


def test_schedule_buffers_bit_exact():
    from diffusion.ddpm import LatentDiffusion
    import bench
    v = load_npz("diffusion.npz")
    d = LatentDiffusion(device="cpu", **bench.MODEL_PARAMS)
    sd = d.sampler.state_dict()
    for k in sd:
        assert np.array_equal(sd[k].numpy(), v["sched." + k]), k
    assert "lvlb_weights" not in sd                      # non-persistent (ddpm_sampler.py:96-97)
    assert np.array_equal(d.sampler.lvlb_weights.numpy(), v["sched.lvlb_weights"])
    ds = d.sampler_list["ddim"]
    for S in (10, 50, 250):
        for eta in (0.0, 1.0):
            ds.make_schedule(dict(num_timesteps=S, ddim_eta=eta, alphas_cumprod=d.sampler.alphas_cumprod))
            assert np.array_equal(ds.ddim_timesteps, v[f"ddim{S}.timesteps"])
            assert np.array_equal(np.asarray(ds.ddim_sigmas, dtype=np.float64), v[f"ddim{S}.eta{eta}.sigmas"])
            assert np.array_equal(np.asarray(ds.ddim_alphas, dtype=np.float64), v[f"ddim{S}.eta{eta}.alphas"])


def test_optimizer_chunk_table():
    from sgdm_amd.optim import CHUNK, chunk_table
    assert chunk_table([1, CHUNK, CHUNK + 1, 3 * CHUNK]) == [0, 1, 2, 4, 7]
    assert chunk_table([]) == [0]


@pytest.mark.parametrize("fname", ["r1_bench_c2_f16x3.json", "r2_bench_c2_f16x3_final.json", "r2_bench_c2_f16x3_final_b.json",
                                   "r3_bench_c2_f16x3.json", "r3_bench_c2_f16x3_final.json", "r4_bench_c2_f16x3.json",
                                   "r5_bench_c2_f16x3.json"])
def test_committed_bench_line_has_the_contract_fields(fname):
    """the bench lines committed under profiles/ (produced by bench.py on the MI355X) carry every field of the contract"""
    import json
    line = json.loads(open(os.path.join(ROOT, "profiles", fname)).read().strip().splitlines()[-1])
    if fname.startswith("r5_"):
        # round 5: in-run device calibration, the C3 per-GPU batch, the one-rank RCCL run of the exchange with its overlap record
        r = line["roofline"]
        assert r["device_mfma_tflops"] > 1000 and 0.3 < r["frac_of_device_ceiling"] < 1.0 and r["device_copy_tbps"] > 3
        assert abs(r["device_ceiling_tflops"] * 3 - r["device_mfma_tflops"]) < 1.0
        assert all("frac_of_device_ceiling" in v for v in r["igemm_by_instance"].values())
        assert line["train_step_bs40"]["batch_per_gpu"] == 40 and line["train_step_bs40"]["ms"] > 0
        tr = line["train_step"]
        assert tr["exchange"] is False and tr["exchange_ms"] is None and "RCCL" not in tr["includes"]
        ex = tr["exchange_world1"]
        assert ex["backend"] == "nccl" and ex["world_size"] == 1 and ex["reserved_cus"] == 16 and ex["exchange"] == "forced on one rank"
        assert 0.0 < ex["first_bucket_at_frac_of_backward"] < 1.0 and ex["exposed_exchange_ms"] >= 0 and len(ex["exchange_buckets"]) >= 3
    if fname.startswith("r4_") or fname.startswith("r5_"):
        # round 4: the metric in full beside the K-step figure, the train step's per-kernel record, an honest checksum flag
        ft = line["full_trajectory"]
        assert ft["steps"] == 1000 and ft["images"] == line["n_gpus"] * line["config"]["batch_per_gpu"]
        assert abs(ft["vs_k_step_extrapolation"] - 1.0) < 0.03
        tr = line["train_step"]
        assert tr["grad_checksums_equal"] is None and tr["world_size"] == 1 and tr["reserved_cus"] == 0
        assert {"sgd_wgrad", "sgd_igemm(forward)", "sgd_igemm(dgrad)"} <= set(tr["roofline"]["per_kernel"])
        assert "_pmc_hbm_c2.json" in line["roofline"]["traffic_source"]
    if fname[:3] in ("r3_", "r4_", "r5_"):
        assert "c2_bs80" in line      # (its grad_checksums_equal is trivially true: one rank; null from round 4 on)
    if fname.startswith("r3_"):
        assert "r3_pmc_hbm" in line["roofline"]["traffic_source"]
    if not fname.startswith("r1_"):
        # round 2: the line is self-sufficient (exact-fp32 figure, C5 and C1 legs, instantiation split, traffic label)
        for k in ("f32_exact", "c5", "c1"):
            assert k in line, k
        assert line["f32_exact"]["frac_of_fp32_mfma_peak"] > 0.5 and "train_step" in line["c5"]
        if "igemm_by_instance" in line["roofline"]:        # added with the last line of the round
            assert set(line["roofline"]["igemm_by_instance"]) == {"taps9_conv3x3", "taps1_conv1x1_linear"}
        else:
            assert fname.endswith("_b.json")
        assert "traffic_source" in line["roofline"] and "c1_full" in line["cpu_baseline"]
        assert line["config"]["launch"].startswith("hipGraph")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "train_step"):
        assert k in line, k
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")


def test_library_exports_nothing_undeclared():
    """every `sgd_*` function the shared library exports is declared in include/sgdm_hip.h (no hidden entry points)"""
    import subprocess
    from sgdm_amd import _lib as L
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T sgd_" in ln}
    assert exported == set(L.SIGNATURES), exported ^ set(L.SIGNATURES)


def test_lost_box_corner_contract_matches_reference_pipeline_masks():
    """guidance.lost_box_in_frame (host integer bookkeeping): corners of a LOST box after the reference's RandomScaleCrop
    (two PIL NEAREST resizes + a crop applied to the box MASK) == the bounding box of the mask the reference's own class
    produced (tests/golden/vis.npz, make_golden_vis.py)"""
    import numpy as np
    from conftest import load_npz
    from sgdm_amd.guidance import lost_box_in_frame
    v = load_npz("vis.npz")
    W0, H0 = (int(t) for t in v["guid.box_orig_size"])
    base, S = (int(t) for t in v["guid.box_crop_resize"])
    for b, p, m in zip(v["guid.box_orig"], v["guid.box_scaled_size_crop_xy"], v["guid.box_mask64"]):
        c = lost_box_in_frame([int(t) for t in b], (W0, H0), (int(p[0]), int(p[1])), (int(p[2]), int(p[3])), base, S)
        ref = np.zeros((S, S), dtype=np.float32)
        ref[c[1]:c[3], c[0]:c[2]] = 1
        assert (ref == m[0]).all(), (b, c)


def test_balanced_tail_workspace_layout_never_shares_a_counter_or_a_slab():
    """sgd_igemm_tail_layout (the kernel's own schedule arithmetic, host side): in every launch geometry, the split tiles of
    all XCDs -- whose K split differs when the last XCD holds fewer tiles -- own disjoint arrival counters and disjoint
    producer slabs inside the workspace sgd_igemm_work_bytes() sizes.  (Round 3: ranges sized by the XCD's own split
    overlapped: 1 NaN evaluation in ~20 of the ch=224 model at batch 1, then a hang.)"""
    from sgdm_amd import _lib as L
    lib = L.load()
    slabs_total = (int(lib.sgd_igemm_work_bytes()) - 256 * 8) // (128 * 256 * 4)
    mixed = 0
    for grid in (256, 64, 8):
        out = (ctypes.c_int32 * (4 * grid))()
        for taps in (9, 1):
            for nchunks in (2, 3, 4, 7, 8, 21, 32, 60):
                for total in list(range(1, 700)) + [1280, 1920, 2560]:
                    assert lib.sgd_igemm_tail_layout(total, nchunks, taps, grid, out) == 0
                    tiles = {}
                    for b in range(grid):
                        split, cnt, slab0, ns = out[4 * b:4 * b + 4]
                        if split:
                            assert ns == split - 1 and 0 <= cnt < 256
                            tiles.setdefault((b & 7, cnt), set()).add((split, slab0))
                    counters = [c for (_, c) in tiles]
                    assert len(counters) == len(set(counters)), (grid, taps, nchunks, total, "counter shared between XCDs")
                    used = []
                    for (xcd, cnt), v in tiles.items():
                        assert len(v) == 1, "the blocks of one split tile agree on split and slab"
                        (split, slab0), = v
                        used += list(range(slab0, slab0 + split - 1))
                    assert len(used) == len(set(used)), (grid, taps, nchunks, total, "slab shared")
                    if grid == 256 and used:
                        assert max(used) < slabs_total
                    mixed += len({s for v in tiles.values() for (s, _) in v}) > 1
    assert mixed > 100          # geometries where XCDs of one launch use different splits were actually covered
