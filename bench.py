#!/usr/bin/env python3
"""Benchmark of the FCVSR per-frame forward hot path on MI355X (contract: see the task prompt / DESIGN.md section 6).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path (model(lrs)) over one batch of `--batch` synthetic 7-frame windows of
180x320 LR frames (4x -> 720x1280), inputs resident in HBM before the timed region.  Clips are independent, so ranks
run disjoint clips with no data-path collective ("weak" scaling); value = frames all ranks produced / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}   # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


def conv_flops_live(model_name: str, H: int, W: int) -> float:
    """Live conv FLOPs (2*MAC) per output frame, Y model (552.33 G (S) / 1294.58 G (full) at 180x320)."""
    per_px = {"S": 552.33e9, "full": 1294.58e9}[model_name] / (180 * 320)
    return per_px * H * W


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """CPU threads this process may actually use (cgroup/affinity aware), capped at 16 (the GPU box's per-GPU share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def stub_run(args, world, rank):
    """No-GPU rehearsal of the rank protocol (rendezvous, barrier, max-over-ranks, rank 0 prints one line)."""
    import torch
    import torch.distributed as dist
    if os.environ.get("FCVSR_BENCH_FAIL_RANK") == str(rank):       # test hook: a rank that dies before the rendezvous
        sys.exit(3)
    if world > 1:
        dist.init_process_group(args.backend if args.backend != "nccl" else "gloo")
    x = torch.zeros(args.batch, 7, 1, 8, 8)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = x + 1.0
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # the training sub-record's rank protocol (collective failure agreement) on CPU tensors
    train = None
    if not args.no_train:
        g = torch.zeros(16)

        def coll():
            if world > 1:
                dist.all_reduce(g)
        err, done, _ = guarded_steps(lambda it: g.add_(1.0), coll, (lambda: dist.barrier()) if world > 1 else (lambda: None),
                                     3, world, rank, "cpu", "gloo")
        train = {"error": err, "steps_done": done} if err is not None else {"steps_done": done, "world": world}
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": world * args.batch * args.steps / max(float(t), 1e-9), "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "data": "stub", "scaling": "weak",
                          "train": train}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def visible_gpu_count():
    """GPUs this process tree may use, WITHOUT touching the HIP runtime: the *_VISIBLE_DEVICES lists if set, else the KFD
    topology nodes that have SIMDs (CPU nodes have simd_count 0).  None when neither source exists (no amdgpu driver)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    base = os.environ.get("FCVSR_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology/nodes")
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            with open(os.path.join(base, d, "properties")) as f:
                for ln in f:
                    if ln.startswith("simd_count"):
                        n += int(ln.split()[1]) > 0
                        break
        except (OSError, ValueError, IndexError):
            continue
    return n


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start N fresh rank processes (one per GPU) BEFORE this
    process touches the GPU, relay their output (rank 0 prints the JSON line on stdout) and return non-zero if any fails.
    Same environment contract as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (reference launcher
    shape: mmedit_train/tools/dist_train.sh:10-19)."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies (bad device ordinal, build missing, ...) must not leave the others waiting in a collective until some outer
    # timeout: poll, and on the first failure stop the remaining ranks (exact PIDs of the children started above)
    rc = 0
    alive = dict(enumerate(procs))
    while alive:
        time.sleep(0.2)
        for r, p in list(alive.items()):
            c = p.poll()
            if c is None:
                continue
            del alive[r]
            if c != 0:
                log(f"rank {r} exited with code {c}")
                rc = rc or c
        if rc and alive:
            log(f"stopping the remaining ranks {sorted(alive)}")
            for p in alive.values():
                p.terminate()
            for p in alive.values():
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
    return rc


def build_info():
    """Compiler flags the loaded library was built with (the packed-FP32 guard of fcvsr_amd/build.py is part of the result)."""
    from fcvsr_amd import build as Bd
    return {"flags": " ".join(Bd.FLAGS), "slp_files": sorted(Bd.SLP_FILES), "no_slp_default": True}


def all_ranks_ok(ok: bool, world: int, dev, backend: str) -> bool:
    """Collective agreement on success: MIN over ranks of a 0/1 flag.  Every rank calls it the same number of times whatever
    happened locally, so a local failure never leaves the others waiting inside a later collective."""
    if world <= 1:
        return ok
    import torch
    import torch.distributed as dist
    f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return bool(int(f.item()) == 1)


def guarded_steps(local_fn, collective_fn, after_warmup_fn, nt, world, rank, dev, backend, err=None):
    """Run 1 warm-up + `nt` steps of {local half under try/except; MIN-agree on success; collective half}.  Returns
    (error or None, timed steps completed, perf_counter at the start of the timed steps).  Every rank executes the same sequence
    of collectives whatever fails locally (FCVSR_BENCH_TRAIN_FAIL_RANK=r: test hook, rank r's local half raises at step 1)."""
    fail_rank = os.environ.get("FCVSR_BENCH_TRAIN_FAIL_RANK")
    done, t1 = 0, None
    for it in range(nt + 1):                                                # iteration 0 = warm-up (allocations, capture)
        ok = err is None
        if ok:
            try:
                if fail_rank is not None and int(fail_rank) == rank and it == 1:
                    raise RuntimeError("FCVSR_BENCH_TRAIN_FAIL_RANK test hook")
                local_fn(it)
            except Exception as e:
                ok, err = False, repr(e)[:300]
        if not all_ranks_ok(ok, world, dev, backend):
            err = err or "another rank failed"
            break
        collective_fn()
        if it == 0:
            after_warmup_fn()
            t1 = time.perf_counter()
        else:
            done += 1
    return err, done, t1


def train_record(args, A, dev, world, rank, sd):
    """BASELINE config 3 (sub-record, every rank takes part): FCVSR-S training step - batch 32 clips sharded 8 x 4 (here: 4
    clips of 7x128x128 -> 512x512 per rank, the reference's RandomCrop(128), train_LD_freqCVSR_S_22.py:187), Charbonnier-sum
    loss, Adam, ONE flat-buffer gradient all-reduce per step over RCCL.  Never part of `value`.

    Failure is COLLECTIVE: the local half of every step (set-up, forward, backward) runs under try/except and the ranks agree
    on a MIN-reduced ok flag before anybody enters the gradient all-reduce; one failing rank makes every rank skip the rest
    with the same number of collectives behind it, and the headline line is still printed."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from fcvsr_amd.arch.schema import state_dict_shapes
    from fcvsr_amd.weights import synthetic_state_dict
    err = None
    step = tx = th = None
    try:
        from fcvsr_amd.train import TrainStep
        tm = A.GShiftNet_S()
        tm.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet_S"), gain=0.5), strict=True)
        tm = tm.to(dev)
        tm.train_precision = args.train_precision
        g = torch.Generator().manual_seed(300 + rank)                       # different data per rank, same weights
        tx = torch.rand(4, 7, 1, 128, 128, generator=g).to(dev)
        th = torch.rand(4, 1, 512, 512, generator=g).to(dev)
        step = TrainStep(tm, lr=1e-4, weight_decay=1e-5, use_graph=bool(args.train_graph))
    except Exception as e:
        err = repr(e)[:300]
    nt = 3
    state = {"lv": float("nan")}

    def local(it):
        loss = step.local_backward(tx, th)
        state["lv"] = float(loss.detach())                                  # host sync: the local half is complete

    def after_warmup():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    err, done, t1 = guarded_steps(local, lambda: step.reduce_and_update(), after_warmup, nt, world, rank, dev, args.backend, err)
    lv, sec = state["lv"], float("nan")
    if err is None:
        torch.cuda.synchronize()
        tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev if world > 1 and args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sec = float(tt) / nt
    if err is not None:
        log(f"train sub-record failed on some rank: {err}")
        return {"error": err, "steps_done": done}
    # the collective of the step, timed alone (same flat buffer, HIP events on the current stream)
    ar_ms = None
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        step.allreduce()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            step.allreduce()
        e1.record()
        torch.cuda.synchronize()
        ar_ms = round(e0.elapsed_time(e1) / 3, 3)
    except Exception as e:
        log(f"all-reduce timing skipped: {e!r}")
    # gradients of the timed (16-bit) mode against the exact-f32 mode on the same clips and weights (rank 0's shard)
    grad_err = None
    if rank == 0 and args.train_precision != "f32":
        try:
            def flat_grads(precision, graph):
                m = A.GShiftNet_S()
                m.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet_S"), gain=0.5), strict=True)
                m = m.to(dev)
                m.train_precision = precision
                st = TrainStep(m, use_graph=graph)
                st.local_backward(tx[:2], th[:2])
                torch.cuda.synchronize()
                return [p.grad.detach().double().reshape(-1).clone() for p in st.allreduce.params], st.names
            g16, names = flat_grads(args.train_precision, bool(args.train_graph))
            g32, _ = flat_grads("f32", False)
            a16, a32 = torch.cat(g16), torch.cat(g32)
            per = [float((u - v).abs().max() / v.abs().max().clamp_min(1e-30)) for u, v in zip(g16, g32)]
            worst = int(np.argmax(per))
            grad_err = {"rel_l2_all": float((a16 - a32).norm() / a32.norm()),
                        "max_over_tensors_of_maxabs_over_tensor_max": per[worst], "worst_tensor": names[worst],
                        "clips": 2, "vs": "exact-f32 training mode, same clips and weights"}
        except Exception as e:
            grad_err = {"error": repr(e)[:200]}
    train = {"workload": "FCVSR-S training step: 4 clips of 7x128x128 -> 512x512 per GPU, Charbonnier-sum, Adam, "
                         "flat f32 gradient all-reduce (SUM)", "world": world, "global_batch": 4 * world,
             "conv_precision": args.train_precision + (" forward / input-gradient / weight-gradient on MFMA (f32 accumulate)" if args.train_precision != "f32" else " (exact)"),
             "ms_per_step": round(sec * 1e3, 2), "clips_per_s": round(4 * world / sec, 2),
             "allreduce_bytes": int(step.allreduce.numel * 4), "allreduce_ms": ar_ms, "finite_loss": bool(np.isfinite(lv)),
             "hipgraph": bool(args.train_graph), "grad_rel_err_vs_f32": grad_err}
    log(f"train sub-record: {train}")
    return train


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="7-frame windows per step per GPU (clips are independent)")
    ap.add_argument("--model", choices=["S", "full"], default="S")
    ap.add_argument("--height", type=int, default=180)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--precision", choices=["f32", "bf16", "f16"], default="bf16",
                    help="conv arithmetic: bf16/f16 MFMA operands with f32 accumulate (BASELINE config), or exact f32")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the batch is split over (2: the persistent resident-weight kernels want 8-clip launches; 4 was the round-1 default)")
    ap.add_argument("--trunk16", type=int, default=1, help="1: spatial activations / SCNet trunk stored in the MFMA dtype")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the forward from a captured hipGraph")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="override the HIP device index (rehearsing N ranks on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sub-records (batch-1 latency, full model, f32 mode)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step sub-record (BASELINE config 3)")
    ap.add_argument("--train-precision", choices=["f32", "bf16", "f16"], default="bf16")
    ap.add_argument("--train-graph", type=int, default=1, help="1: replay forward + backward of the training step from a hipGraph")
    ap.add_argument("--stub", action="store_true",
                    help="launch-plumbing rehearsal without a GPU: the step is a no-op on CPU tensors and the line says so "
                         "(metric 'stub'); used by tests/test_dist_cpu.py with --backend gloo")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # The launcher parent imports nothing that could open /dev/kfd (no torch, no HIP): a process that has initialised the
        # GPU must not fork + exec on this pool.  Devices are counted from the environment / sysfs; when neither can tell, the
        # children validate their own ordinal (LOCAL_RANK < device_count()) and the first failure stops the rest.
        if args.device is None and (not args.stub or "FCVSR_KFD_TOPOLOGY" in os.environ):   # a stub rehearsal needs no GPU
            have = visible_gpu_count()
            if have is not None and have < args.gpus:
                raise SystemExit(f"--gpus {args.gpus} but only {have} HIP device(s) are visible "
                                 f"(rehearse N ranks on one GPU with --device 0)")
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N")
    if args.stub:
        return stub_run(args, world, rank)
    local = local if args.device is None else args.device
    if local >= torch.cuda.device_count():
        log(f"rank {rank}: device ordinal {local} but only {torch.cuda.device_count()} HIP device(s) are visible")
        sys.exit(4)                                       # the launcher stops the other ranks (self_launch / torchrun)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from fcvsr_amd import hip
    from fcvsr_amd.arch import CVSR_freq as A
    from fcvsr_amd.arch.schema import state_dict_shapes
    from fcvsr_amd.weights import synthetic_state_dict

    ctor = "GShiftNet_S" if args.model == "S" else "GShiftNet"
    sd = synthetic_state_dict(state_dict_shapes(ctor), gain=0.5)
    model = getattr(A, ctor)()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    model.precision = args.precision
    model.streams = args.streams
    model.use_graph = bool(args.graph)
    model.trunk16 = bool(args.trunk16)
    B, H, W = args.batch, args.height, args.width
    rs = np.random.RandomState(1 + rank)                      # different clips per rank, same weights
    x = torch.from_numpy(rs.rand(B, 7, 1, H, W).astype(np.float32)).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"model {ctor} on {dev}, input {tuple(x.shape)}; warmup {args.warmup}")
    with torch.no_grad():
        for _ in range(args.warmup):
            y = model(x)
            torch.cuda.synchronize()
            log("warmup step done")
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = model(x)
        barrier()
        dt = time.perf_counter() - t0
    per_rank_dt = [dt]
    if world > 1:
        cdev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)                              # every rank's own time: per-GPU fps in the line
        per_rank_dt = [float(v.item()) for v in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(y).all())
    log(f"timed {args.steps} steps in {dt:.3f} s")
    frames = world * B * args.steps
    fps = frames / dt

    roofline = None
    timed_equals_eager = None
    if not args.no_roofline:
        # Dominant kernel class = the convolution kernel (>= 98 % of FLOPs).  One extra instrumented step: every conv launch
        # is bracketed by HIP events on the launch stream; achieved = sum(algorithmic FLOPs) / sum(durations).
        hip.PROFILE = []
        model.streams, model.use_graph = 1, False          # one stream, eager launches: isolated per-launch durations
        with torch.no_grad():
            y_eager = model(x)
        torch.cuda.synchronize()
        model.streams, model.use_graph = args.streams, bool(args.graph)
        # the timed configuration (hipGraph replay, sub-batches on several streams) must reproduce the eager single-stream
        # launches bit for bit: clips are independent and every reduction of the path has a fixed order
        timed_equals_eager = bool(torch.equal(y, y_eager))
        assert timed_equals_eager, (f"timed configuration differs from the eager single-stream forward: max-abs "
                                    f"{float((y - y_eager).abs().max()):.3e}")
        recs = hip.PROFILE
        hip.PROFILE = None
        # Dominant kernel = the convolution kernel with the largest share of the step's GPU time.  Every record carries the
        # name of the kernel the dispatcher really launched (fcvsr_last_conv_kernel), so this follows dispatcher changes:
        # the LDS-resident-weight 3x3 kernel conv3_res_kernel<BF16, DST16, NCH> for the 64 / 128-channel layers at batch
        # sizes that amortise its weight copy, conv3_lean_kernel otherwise; conv_direct_kernel in the exact-f32 mode.
        cls = [r for r in recs if r[3] == ("direct" if args.precision == "f32" else "mfma")]
        by_kernel = {}
        for r in cls:
            kn = r[6] if r[3] == "mfma" else "conv_direct_kernel"
            by_kernel.setdefault(kn, []).append(r)
        kernel_ms = {kn: sum(r[0].elapsed_time(r[1]) for r in rs) for kn, rs in by_kernel.items()}
        var = max(kernel_ms, key=kernel_ms.get)
        dom = by_kernel[var]
        tot_ms = sum(r[0].elapsed_time(r[1]) for r in dom)
        tot_fl = sum(r[2] for r in dom)
        cls_ms = sum(r[0].elapsed_time(r[1]) for r in cls)
        cls_fl = sum(r[2] for r in cls)
        peak = PEAK_TFLOPS[args.precision]
        ach = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        traffic = None
        try:        # HBM bytes per launch from the committed PMC passes (same workload), see profiles/pmc_traffic.json
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f)
            c = pm["config"]
            if (c["model"], c["batch"], c["precision"], c["height"], c["width"], c.get("act16", False)) == \
                    (args.model, B, args.precision, H, W, bool(args.trunk16)) and pm.get("kernel", "").replace("void fcvsr::", "").startswith(var.rstrip(">")):
                traffic = round(pm["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
        n_dom = max(1, len(dom))
        roofline = {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": round(sum(r[5] for r in dom) / n_dom),
                    "algorithmic_flops_per_launch": round(tot_fl / n_dom),
                    "kernel": var,
                    "launches_per_step": len(dom), "avg_launch_us": round(tot_ms * 1e3 / n_dom, 2),
                    "kernel_ms_per_step": round(tot_ms, 3),
                    "how": "one extra single-stream eager step, HIP events around every launch on the launch stream",
                    "conv_kernels_ms_per_step": {kn: round(v, 3) for kn, v in sorted(kernel_ms.items(), key=lambda kv: -kv[1])},
                    "all_conv_kernels": {"tflops": round(cls_fl / (cls_ms * 1e-3) / 1e12, 3) if cls_ms > 0 else 0.0,
                                         "launches_per_step": len(cls), "ms_per_step": round(cls_ms, 3)}}
        if args.precision != "f32":
            # Context, not the price: what the vendor's own GEMM (hipBLASLt through torch.matmul, 8192^3 in the same 16-bit type)
            # sustains on THIS box at its power limit.  `peak` / `frac` above stay on the data-sheet figure.
            try:
                mdt = torch.bfloat16 if args.precision == "bf16" else torch.float16
                ga = torch.randn(8192, 8192, device=dev, dtype=mdt)
                gb = torch.randn(8192, 8192, device=dev, dtype=mdt)
                gc = torch.empty(8192, 8192, device=dev, dtype=mdt)
                for _ in range(3):
                    torch.matmul(ga, gb, out=gc)
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                for _ in range(10):
                    torch.matmul(ga, gb, out=gc)
                g1.record()
                torch.cuda.synchronize()
                vt = 10 * 2.0 * 8192.0 ** 3 / (g0.elapsed_time(g1) * 1e-3) / 1e12
                roofline["vendor_gemm"] = {"what": "hipBLASLt 8192x8192x8192 via torch.matmul, same box, same dtype", "tflops": round(vt, 1),
                                           "frac_of_peak": round(vt / peak, 4), "achieved_over_vendor_gemm": round(ach / vt, 4)}
                del ga, gb, gc
            except Exception as e:                             # calibration only: never fails the bench
                roofline["vendor_gemm"] = {"error": repr(e)[:200]}

    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import fcvsr_oracle as O                  # checker + CPU baseline leg only (never the timed path)
        ncpu = host_threads()
        torch.set_num_threads(ncpu)
        log(f"cpu baseline: oracle on {ncpu} threads")
        xc = x[:1].cpu()
        with torch.no_grad():
            ref0 = O.forward(sd, xc)
            ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                O.forward(sd, xc)
                ts.append(time.perf_counter() - t1)
                log(f"cpu baseline run {ts[-1]:.2f} s")
        cpu = {"value": round(1.0 / float(np.median(ts)), 4), "unit": "frames/s", "cores": torch.get_num_threads(),
               "kind": "port", "sample": f"3 forwards of one {H}x{W} 7-frame window (median), fp32 torch CPU oracle"}
        # the PSNR half of the metric: the TIMED output of clip 0 against the CPU oracle (reference arithmetic in fp32) on
        # the same input and weights; [0,255] scale, formula of CVSR_train/metric/psnr_ssim.py:314-317
        d = (y[:1].detach().cpu().double() - ref0.double())
        mse = float((d * 255.0).pow(2).mean())
        parity = {"psnr_vs_oracle_db": round(float(20 * np.log10(255.0 / np.sqrt(mse))) if mse > 0 else 999.0, 3),
                  "max_abs": float(d.abs().max()), "clip": 0, "scale": "[0,1] for max_abs, [0,255] for PSNR",
                  "timed_equals_eager_single_stream_bitwise": timed_equals_eager}
        log(f"parity vs oracle: {parity}")

    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}

        def timeit(mdl, xin, nw, nt):
            with torch.no_grad():
                for _ in range(nw):
                    mdl(xin)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(nt):
                    o = mdl(xin)
                torch.cuda.synchronize()
            assert bool(torch.isfinite(o).all())
            return (time.perf_counter() - t1) / nt

        # latency of ONE window (the reference's own FPS script times batch 1, test_LD_freqCVSR_S_FPS.py:62-74)
        model.streams = 1
        extras["batch1_ms"] = round(timeit(model, x[:1], 3, 20) * 1e3, 3)
        # exact-f32 mode of the same model (the mode that meets the 1e-4 max-abs gate)
        model.precision = "f32"
        extras["f32_fps"] = round(B / timeit(model, x, 1, 2), 2)
        with torch.no_grad():
            yf = model(x[:1])
        if not args.no_cpu_baseline:
            extras["f32_max_abs_vs_oracle"] = float((yf.cpu() - ref0).abs().max())
        model.precision, model.streams = args.precision, args.streams
        del yf
        fm = None
        if args.model == "S":
            # the full model (BASELINE configs 2 / 4) with the same settings
            log("extras: full model")
            fm = A.GShiftNet()
            fm.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet"), gain=0.5), strict=True)
            fm = fm.to(dev)
            fm.precision, fm.streams, fm.use_graph, fm.trunk16 = args.precision, args.streams, bool(args.graph), bool(args.trunk16)
            extras["full_model_fps"] = round(B / timeit(fm, x, 1, 3), 2)
        # BASELINE config 5 as a number: REDS4-shaped sequences (4 x 100 LR frames of 180x320, anna_file/REDS4_GT.txt) streamed
        # from PINNED HOST memory through the harness scheduler - every LR frame uploaded once, windows gathered on the device,
        # uint8 SR frames copied back to pinned host memory: the PCIe legs are INSIDE this number (they never are in `value`)
        if (H, W) == (180, 320):
            try:
                from fcvsr_amd.harness.infer import StreamedSuperResolver
                gs = torch.Generator().manual_seed(55)
                seqs = [torch.rand(100, 1, H, W, generator=gs).pin_memory() for _ in range(4)]
                stream = {"workload": "4 sequences x 100 LR frames 180x320 in pinned host memory -> uint8 720x1280 frames in pinned host "
                                      "memory, replicate-padded 7-frame windows, batch %d" % B}
                for name, mdl, resident in (("S", model, fps), ("full", fm, extras.get("full_model_fps"))):
                    if mdl is None:
                        continue
                    ssr = StreamedSuperResolver(mdl, batch=B)
                    ssr.run(seqs)                                     # warm-up: buffers page-locked, hipGraph captured
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    out = ssr.run(seqs)
                    sec = time.perf_counter() - t1
                    assert sum(v[1].shape[0] for v in out.values()) == 400
                    stream[name] = {"fps": round(400 / sec, 2), "resident_input_fps": resident,
                                    "ratio": round(400 / sec / resident, 4) if resident else None,
                                    "h2d_bytes_per_frame": ssr.stats["h2d_bytes"] // 400, "d2h_bytes_per_frame": ssr.stats["d2h_bytes"] // 400,
                                    "frames_uploaded": ssr.stats["frames_uploaded"]}
                    del ssr
                extras["stream"] = stream
                extras["stream_fps"] = stream.get("S", {}).get("fps")
                del seqs
            except Exception as e:
                extras["stream"] = {"error": repr(e)[:300]}
        del fm
        log(f"extras: {extras}")

    # BASELINE config 3 (sub-record, every rank takes part): FCVSR-S training step - batch 32 clips sharded 8 x 4 (here: 4 clips
    # of 7x128x128 -> 512x512 per rank, the reference's RandomCrop(128), train_LD_freqCVSR_S_22.py:187), Charbonnier-sum loss,
    # Adam, ONE flat-buffer gradient all-reduce per step over RCCL.  Never part of `value`.
    train = None
    if not args.no_train and not args.stub:
        train = train_record(args, A, dev, world, rank, sd)

    if rank == 0:
        line = {
            "metric": "SR frames/sec (7-frame window, 4x 180x320->720x1280) + PSNR vs ref", "value": round(fps, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"FCVSR-{args.model} 4x inference, {B}x7x{H}x{W} -> {4*H}x{4*W} synthetic clips, "
                                   f"random-init (key-seeded) weights", "batch_per_gpu": B, "streams_per_gpu": args.streams, "hipgraph": bool(args.graph), "act16": bool(args.trunk16),
                       "parallelism": f"clip-dp{world}"},
            "frames_per_sec_per_gpu": round(fps / world, 3),
            "conv_tflops_end_to_end": round(fps * conv_flops_live(args.model, H, W) / 1e12, 3),
            "per_gpu_fps": [round(B * args.steps / t_r, 3) for t_r in per_rank_dt],
            "dist": {"world": world, "backend": (dist.get_backend() if world > 1 else None),
                     "rccl_world": (dist.get_world_size() if world > 1 and dist.get_backend() == "nccl" else None)},
            "build": build_info(),
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "extras": extras, "train": train,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
