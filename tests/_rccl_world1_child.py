"""Child process of tests/test_hip_rccl_world1.py: a world-size-1 ``nccl`` (= RCCL) process group on the one GPU.

Runs one C2 training step at B = 16 twice on the same model and inputs:
  A. the plain single-process backward (``hip_ddp = False``), with the CU reserve the exchange would use, and
  B. the data-parallel backward with the exchange FORCED (``hip_force_exchange``): gradient arena, bucketed all-reduce
     through ProcessGroupNCCL on the side stream, event ordering, CU reserve from the policy, the exchange's own
     communicator with its workgroup cap, initial-state broadcast.
Prints one JSON line; the parent asserts on it.  (A fresh process: RCCL reads its environment once per process.)
"""
import json
import os
import socket
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-guided-diffusion-models_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    import bench
    from sgdm_amd import ddp
    from sgdm_amd.diffusion import LatentDiffusion
    from sgdm_amd.synth import synth_batch
    torch.cuda.set_device(0)
    os.environ.pop("NCCL_MAX_NCHANNELS", None)
    out = dict(workload=workload, batch=B)
    with tempfile.TemporaryDirectory() as td:
        dist.init_process_group("nccl", init_method=f"file://{os.path.join(td, 'store')}", rank=0, world_size=1)
        out["backend"] = str(dist.get_backend())
        out["comm_before"] = ddp.communicator_exists()
        wl = bench.WORKLOADS[workload]
        model, _, _ = bench.build_model(wl, torch.device("cuda"), "f16x3", B)
        model.dropout = 0.0
        model.train()
        diff = LatentDiffusion(device="cuda", **bench.MODEL_PARAMS).train()
        diff.set_denoise_fn(model.forward, model.forward_with_cond_scale)
        data = synth_batch(wl["method"], B, 64, wl["cond_dim"], wl["layout_dim"], seed=5)
        g = torch.Generator().manual_seed(5)
        t = torch.randint(0, 1000, (B,), generator=g).cuda()
        noise = torch.randn(B, 3, 64, 64, generator=g).cuda()
        mask = (torch.rand(B, generator=g) < 0.1).cuda()
        x0 = data["image"].cuda()
        cond = data["cond"].cuda() if wl["kind"] == "unet_fast" else data["cond"].float().cuda()
        layout = data["layout"].cuda() if "layout" in data else None

        def step():
            for p in model.parameters():
                p.grad = None
            loss, _ = diff.p_losses(x0, t, noise, cond=cond, layout=layout, cond_drop_prob=0.1, cond_drop_mask=mask)
            loss.backward()
            torch.cuda.synchronize()
            return float(loss.detach())

        # ---- A: no exchange, the reserve by attribute on EVERY backward launch; B below runs with the reserve windows off as
        # well (same grids as A, so the bits must agree); the windows come on for the third step
        os.environ["SGDM_RESERVE_WINDOWS"] = "0"
        # ---- A
        model.hip_ddp = False
        model.hip_reserve_cus = ddp.reserve_setting()
        loss_a = step()
        ref = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        eng = next(iter(model._engines.values()))
        out["plain_backward_has_arena"] = eng.backward.arena is not None
        # ---- B: the exchange forced through the one-rank RCCL group
        for e in model._engines.values():
            e.backward = None
        del model.hip_reserve_cus
        model.hip_ddp = True
        model.hip_force_exchange = True
        out["reserved_cus"] = ddp.reserved_cus(model)
        out["sent"] = ddp.sync_initial_state(model)                # broadcast through RCCL (one rank: values unchanged)
        out["nchannels_env"] = os.environ.get("NCCL_MAX_NCHANNELS")
        t0 = time.time()
        loss_b = step()
        out["step_seconds"] = round(time.time() - t0, 2)
        red = eng.backward.reducer
        out["arena_buckets"] = len(eng.backward.arena.buckets)
        out["reducer_active"] = bool(red.active)
        out["stream_ordered"] = bool(red.stream_ordered)
        out["own_group"] = red.group is not None
        out["grid_cap"] = int(eng._grid_cap)
        out["cus"] = int(torch.cuda.get_device_properties(0).multi_processor_count)
        out["backward_grid_caps"] = sorted({int(a.grid_cap) for a, _ in eng.backward.late})
        out["forward_grid_caps"] = sorted({int(a.grid_cap) for a, _ in eng._late})
        out["overlap"] = red.overlap_stats()
        out["comm_after"] = ddp.communicator_exists()
        got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        out["n_grads"] = len(got)
        out["same_keys"] = sorted(got) == sorted(ref)
        bad = [k for k in ref if k not in got or not torch.equal(got[k], ref[k])]
        out["mismatched"] = bad[:8]
        out["losses_equal"] = loss_a == loss_b
        out["aliased_into_arena"] = sum(1 for k, p in model.named_parameters()
                                        if p.grad is not None and eng.backward.arena.flat.data_ptr() <= p.grad.data_ptr()
                                        < eng.backward.arena.flat.data_ptr() + eng.backward.arena.flat.numel() * 4)
        # a second step re-uses communicator, arena and marks; its record is the steady state (the first step's holds the
        # creation of the exchange's communicator: tens of ms of host time inside the backward)
        step()
        out["overlap_second_step"] = red.overlap_stats()
        bad2 = [k for k in ref if not torch.equal(model.get_parameter(k).grad, ref[k])]
        out["second_step_mismatched"] = bad2[:8]
        # third step: the reserve only inside the windows behind each bucket's enqueue -- most backward launches run on the
        # whole device, so tiles of their last rounds split differently along K: same gradients up to the rounding of a sum
        os.environ["SGDM_RESERVE_WINDOWS"] = "1"
        eng.backward.apply_grid_cap()
        step()
        caps = [int(a.grid_cap) for a, _ in eng.backward.late]
        out["reserve_windows"] = dict(capped=sum(1 for c_ in caps if c_ > 0), launches=len(caps),
                                      entries=list(eng.backward._windows or ()))
        worst = 0.0
        for k in ref:
            gk = model.get_parameter(k).grad
            den = float(ref[k].abs().max())
            if den > 0:
                worst = max(worst, float((gk - ref[k]).abs().max()) / den)
        out["third_step_max_rel"] = worst
        # ---- C: the exchange entry of the C-ABI (include/sgdm_hip.h: sgd_allreduce_bucket, SURVEY 8(b) last row) on a communicator
        # made with librccl's OWN API -- what a host that does not go through torch.distributed would do: event behind the
        # producer, side stream waits, in-place SUM of the bucket, compute stream joins
        import ctypes as C
        from sgdm_amd import _lib as L
        rccl_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        rccl = C.CDLL(rccl_path)

        class UID(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid, comm = UID(), C.c_void_p()
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        out["cabi_comm_rc"] = [int(rccl.ncclGetUniqueId(C.byref(uid))), int(rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0))]
        lib = L.load()
        out["cabi_bind_rc"] = int(lib.sgd_exchange_bind(rccl_path.encode()))
        buf = torch.randn(1 << 22, device="cuda")
        want = buf.clone()
        side, ev = torch.cuda.Stream(), torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            out["cabi_allreduce_rc"] = int(lib.sgd_allreduce_bucket(comm, C.c_void_p(buf.data_ptr()), buf.numel(), side.cuda_stream))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        out["cabi_allreduce_equal"] = bool(torch.equal(buf, want))           # SUM over one rank
        out["cabi_bad_args_rc"] = int(lib.sgd_allreduce_bucket(None, C.c_void_p(buf.data_ptr()), 4, None))
        rccl.ncclCommDestroy(comm)
        dist.barrier()
        dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
