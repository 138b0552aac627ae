#!/usr/bin/env python3
"""Train-step benchmark of the DualQ-SELD-TCN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c1|c4|c5] [--mode graph|eager]

One step = zero_grad -> forward -> BCE + 5*MSE -> backward -> (gradient exchange) -> Adam on one synthetic
minibatch resident in HBM.  Prints ONE JSON line (rank 0) with the whole-job samples/s, the
roofline of the dominant kernel (and of the HBM-bound first-layer DualQ-Conv the north-star names) measured
with HIP events, and the CPU baseline (the oracle timed on the host cores, rank 0, N = 1 only).

`--mode graph` (default) replays the step as recorded HIP graphs (train.GraphedTrainStep): the same kernels,
no host time between them.  HIP events cannot bracket a kernel inside a replayed graph, so the per-kernel
roofline figures come from an instrumented EAGER pass of the same step run right after the timed region, in
this process, on the stream the kernels are launched on (`roofline.measured_over` says so).

N > 1: one process per GPU, two RCCL all-reduces of the flat gradient buffer per step (dp.BucketedGradSync), weak
scaling (per-GPU batch fixed: the workload's own per-GPU batch, e.g. 16 for the 4- and 8-GPU configs c4 / c5).
Either launch it under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the ranks find
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or plainly as `python bench.py --gpus N`: the process
then is a PARENT that never touches the GPU -- it starts that same torch.distributed.run command as a child on a free
port of 127.0.0.1, relays rank 0's JSON line and exits with the child's code.
"""
import argparse
import importlib
import json
import os
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "sound-event-localization-and-detection_amd"

# BASELINE.json configs at the synthetic benchmark shape F=128, T=512 (F=128 needs U = L)
WORKLOADS = {
    "c1": dict(name="SELD-TCN-S1-PHI_8ch (real)", domain="R", domain_classifier="R", input_channels=8, batch=2,
               cnn_filters=[64] * 3, G=128, U=64, V=[128, 128], fc_layers=[128]),
    "c2": dict(name="QSELD-TCN-S1-PHI_parallel_8ch", domain="Q", domain_classifier="R", input_channels=8, batch=32,
               cnn_filters=[64] * 3, G=128, U=64, V=[128, 128], fc_layers=[128]),
    "c3": dict(name="DQSELD-TCN-S1-PHI_8ch", domain="DQ", domain_classifier="DQ", input_channels=8, batch=32,
               cnn_filters=[192] * 3, G=384, U=192, V=[384, 384], fc_layers=[384]),
    # configs 4 and 5 are quoted over 4 / 8 GPUs at 16 samples per GPU
    "c4": dict(name="DQSELD-TCN-S1-PHI_16chMagPhase", domain="DQ", domain_classifier="DQ", input_channels=16, batch=16,
               cnn_filters=[192] * 3, G=384, U=192, V=[384, 384], fc_layers=[384]),
    "c5": dict(name="DQSELD-TCN micAMagPhaseParallelmicBMagPhase two-stream", domain="DQ", domain_classifier="R",
               input_channels=16, batch=16, cnn_filters=[192] * 3, G=384, U=192, V=[384, 384], fc_layers=[128],
               parallel_ConvTC_block="2Parallel", parallel_magphase=True),
}
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_FP32_MFMA_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32, dense
# cnn.0 forward, the one HBM-bound DualQ-Conv: the fast-product kernel with 1 or 2 block channels per K chunk
# (only the networks' first layers have so few), or the block-matrix short-K kernel when that path is forced
FIRST_LAYER_KERNELS = ("first_stage_fwd", "hcq_first_pool_kernel", "hcq_first_kernel", "hcq_conv_kernel<3, 3, 1,", "hcq_conv_kernel<3, 3, 2,", "hc_conv_smallk_kernel")


def first_layer_label(labels):
    return next((k for k in labels if k.startswith(FIRST_LAYER_KERNELS)), None)


def model_kwargs(w, freq=128, time_dim=512):
    """freq = 128 is the shape BASELINE.json's metric is quoted on (U = L); freq = 256 is the config-exact
    input (App. C G3), where the TCN is twice as wide (L = 2 x filters), so U doubles with it."""
    w = dict(w, U=w["U"] * (freq // 128))
    return dict(time_dim=time_dim, freq_dim=freq, input_channels=w["input_channels"], output_classes=14,
                domain=w["domain"], domain_classifier=w["domain_classifier"], cnn_filters=w["cnn_filters"],
                kernel_size_cnn_blocks=3, pool_size=[[8, 2], [8, 2], [2, 2]], pool_time="TCN", D=[10],
                dilation_mode="fibonacci", G=w["G"], U=w["U"], kernel_size_dilated_conv=3, spatial_dropout_rate=0.5,
                V=w["V"], V_kernel_size=3, fc_layers=w["fc_layers"], fc_activations="linear", fc_dropout="Last",
                dropout_perc=0.3, class_overlaps=3, use_bias_conv=0, use_bias_linear=1, batch_norm="BN",
                parallel_ConvTC_block=w.get("parallel_ConvTC_block", "False"),
                parallel_magphase=w.get("parallel_magphase", False))


def pmc_traffic(kernel_label):
    """HBM-side bytes per launch of `kernel_label` from the committed counter passes of this same command
    (profiles/*pmc_traffic.json, written by tools/pmc_traffic.py from separate `rocprofv3 --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` runs, FETCH_SIZE doubled as the gfx950 guide prescribes), or None if the kernel was not
    sampled.  Counters cannot be read from inside this process: the figure is per launch of the same kernel on the
    same workload, the file it came from is named in `traffic_source`."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, None
    try:
        kernels = json.load(open(files[-1]))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    if kernel_label.startswith("first_stage_fwd"):
        # the stage forward is several launches (csrc/first_stage.hip): input second moments, their folds, BatchNorm from them,
        # the finishing pooling convolution -- its traffic is their sum
        parts = ("fs_gram_kernel", "fs_sum_parts_kernel", "fs_gram_fold_kernel", "fs_bn_from_gram_kernel", "hcq_first_pool_kernel")
        hits = [v for name, v in kernels.items() if name.startswith(parts)]
        if not any(name.startswith("hcq_first_pool_kernel") for name in kernels):
            return None, os.path.relpath(files[-1], ROOT)
        return round(sum(v["traffic_bytes"] for v in hits)), os.path.relpath(files[-1], ROOT)
    k = kernels.get(kernel_label)
    if k is None:          # the timer's label may carry fewer template arguments than the profiler's symbol
        stem = kernel_label.rstrip(">")
        hits = [v for name, v in kernels.items() if name.startswith(stem + ",") or name.startswith(stem + ">")]
        k = hits[0] if len(hits) == 1 else None
    return (round(k["traffic_bytes"]) if k else None), os.path.relpath(files[-1], ROOT)


def _cpu_model():
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line.startswith("Model name"):
                return line.split(":", 1)[1].strip()
    except (OSError, subprocess.SubprocessError):
        pass
    return "unknown"


def cpu_baseline(w, freq=128, seconds_budget=14.0):
    """The oracle (CPU restatement of the reference's algorithm, `assembled` = one real conv per layer exactly as
    quaternion_ops.py:125-147 does) timed on this host: same step definition, dropout on, B = min(B, 8), at all host
    threads and at 8; a bounded sample: one warm-up step, then the median of up to 5 steps within
    `seconds_budget` per thread setting."""
    from oracle import seld_oracle as O
    import numpy as np
    pkg = importlib.import_module(PKG)
    batch = min(w["batch"], 8)
    kw = model_kwargs(w, freq)
    np.random.seed(1)
    torch.manual_seed(1)
    m = pkg.model.SELD_Model(**kw)           # host-side construction only (weights); never run on the CPU
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    leaves = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    cfg = O.SeldConfig(**kw)
    opt = torch.optim.Adam(leaves, lr=1e-4)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch, w["input_channels"], freq, 512, generator=g)
    target = torch.cat(((torch.rand(batch, 64, 42, generator=g) < 0.1).float(),
                        torch.rand(batch, 64, 126, generator=g) * 2 - 1), 2)

    def step():
        opt.zero_grad()
        sed, doa = O.seld_forward(sd, cfg, x, train=True, mode="assembled", dropout=True)
        loss = O.seld_loss(sed, doa, target, 42)
        loss.backward()
        opt.step()

    all_threads = torch.get_num_threads()
    runs = []
    for threads in sorted({all_threads, min(8, all_threads)}, reverse=True):
        torch.set_num_threads(threads)
        step()                                # warm-up
        times, t_begin = [], time.time()
        while len(times) < 2 or (time.time() - t_begin < seconds_budget and len(times) < 5):
            t0 = time.time()
            step()
            times.append(time.time() - t0)
        runs.append(dict(threads=threads, value=round(batch / statistics.median(times), 3), steps=len(times)))
    torch.set_num_threads(all_threads)
    best = max(runs, key=lambda r: r["value"])
    return dict(value=best["value"], unit="samples/s", cores=best["threads"], kind="port", cpu=_cpu_model(),
                host_threads=all_threads, runs=runs,
                sample=f"median of {best['steps']} train steps (after 1 warm-up) of the same workload at batch {batch} "
                       f"(oracle, torch-CPU fp32, dropout on); `runs` lists every thread setting, `value` is the faster")


def launch_ranks(argv, n):
    """`python bench.py --gpus N` without a torch.distributed environment: start the N ranks as children of THIS process
    (which has not touched the GPU and never will: no exec, no HIP call), relay rank 0's JSON line, return the exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SELD_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:                    # rank 0 prints the one JSON line; anything else on stdout goes to stderr
        txt = out.strip()
        if line is None and txt.startswith("{") and '"metric"' in txt:
            line = txt
        elif txt:
            print(txt, file=sys.stderr)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench: the ranks exited cleanly but printed no result line", file=sys.stderr)
        rc = 1
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--freq", type=int, default=128, choices=[128, 256],
                    help="frequency bins of the input: 128 = the shape the metric is quoted on, 256 = config-exact")
    ap.add_argument("--mode", default=os.environ.get("SELD_BENCH_MODE", "auto"), choices=["auto", "graph", "eager"],
                    help="how the step is issued: one recorded HIP graph per step, eager launches, or (auto) whichever of "
                         "the two was faster over a few untimed steps after the warm-up -- same kernels, same work")
    ap.add_argument("--roofline-steps", type=int, default=5, help="instrumented eager steps after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if os.environ.get("SELD_BENCH_CHILD"):
            raise SystemExit("bench.py: rank process without WORLD_SIZE in its environment")
        raise SystemExit(launch_ranks(sys.argv[1:], args.gpus))          # before anything touches the GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    pkg = importlib.import_module(PKG)
    H, T, DP = pkg.hip_ops, pkg.train, pkg.dp
    # Rehearsal hooks for a one-GPU box (never set by the driver): SELD_BENCH_BACKEND=gloo with
    # SELD_BENCH_SINGLE_DEVICE=1 runs every rank on cuda:0 to exercise the multi-process path without RCCL.
    rank, local, world = DP.init_from_env(os.environ.get("SELD_BENCH_BACKEND", "nccl"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("SELD_BENCH_SINGLE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    w = WORKLOADS[args.workload]
    batch = args.batch or w["batch"]
    import numpy as np
    np.random.seed(1)
    torch.manual_seed(1)
    DP.seed_rank_streams(rank)
    model = pkg.model.SELD_Model(**model_kwargs(w, args.freq)).to(dev).train()
    opt = T.FlatAdam(model.parameters(), lr=1e-4, late=DP.late_parameters(model) if world > 1 else None)
    DP.broadcast_parameters(opt.flat_param)
    sync = DP.BucketedGradSync(opt, model)
    x, target = T.synthetic_batch(batch, w["input_channels"], args.freq, 512, 42, 1234 + rank, dev)

    def eager_step():
        return DP.dp_train_step(model, opt, sync, x, target, 42, T.seld_loss_fn)

    # ---- warm-up = per-kernel survey: every conv launch of steps 1..W-1 is bracketed by HIP events (step 0 is cold:
    # module load, allocator growth) to find the dominant kernel.  The survey runs with the weight-gradient side
    # stream off: kernels that overlap on two queues stretch each other's brackets, and the survey is about which
    # kernel costs most by itself.
    survey = {}
    side_env = os.environ.get("SELD_WGRAD_SIDE_STREAM")
    timing = not args.no_kernel_timer
    n_warm = max(args.warmup, 2 if timing else 0)
    for i in range(n_warm):
        if timing and i == 1:
            torch.cuda.synchronize()
            H.kernel_timer.reset()
            H.kernel_timer.only = None
            H.kernel_timer.active = True
            os.environ["SELD_WGRAD_SIDE_STREAM"] = "0"
        eager_step()
    torch.cuda.synchronize()
    if timing:
        H.kernel_timer.active = False
        survey = H.kernel_timer.summary()
        H.kernel_timer.reset()
        if side_env is None:
            os.environ.pop("SELD_WGRAD_SIDE_STREAM", None)
        else:
            os.environ["SELD_WGRAD_SIDE_STREAM"] = side_env
    dominant = max(survey, key=lambda k: survey[k]["ms"]) if survey else None

    step_mode = "eager" if args.mode == "eager" else "graph"
    step = eager_step
    if args.mode in ("graph", "auto"):
        try:
            runner = T.GraphedTrainStep(model, opt, x, target, 42, 1.0, 5.0, sync=sync, warmup=1)
            runner()                # one replay outside the timed region
            step = runner
        except Exception as exc:    # never lose the measurement to a capture problem: the eager step is the same work
            torch.cuda.synchronize()
            step_mode = f"eager (graph recording failed: {type(exc).__name__}: {str(exc)[:120]})"
            if rank == 0:
                print("bench: " + step_mode, file=sys.stderr)
    if world > 1:
        # a recording that failed on ONE rank must not leave the ranks on different step sequences (their collectives
        # would no longer pair up): everybody falls back together
        ok = torch.tensor([1.0 if step is not eager_step or args.mode == "eager" else 0.0], device=dev)
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        if ok.item() < 0.5 and step is not eager_step:
            step, step_mode = eager_step, "eager (graph recording failed on another rank)"
    if args.mode == "auto" and step is not eager_step:
        # both issue the same kernels; which one is faster depends on how fast this host launches (eager) against
        # what the graph replay costs: 6 untimed steps each, rank 0 decides for everybody
        def probe(fn, n=6):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        t_graph, t_eager = probe(step), probe(eager_step)
        pick = torch.tensor([1.0 if t_eager < t_graph else 0.0], device=dev)
        if world > 1:
            torch.distributed.broadcast(pick, 0)
        if pick.item() > 0.5:
            step, step_mode = eager_step, "eager"
        step_mode += f" (auto: probe {t_graph * 1e3:.2f} ms recorded / {t_eager * 1e3:.2f} ms eager per step)"
        # the probes leave the OTHER mode's host state behind (the first replay after eager steps brings the device-resident
        # step state and the packed-form table up to date: 30-70 ms once): one untimed step of the chosen mode settles it
        step()
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps between barrier + synchronize; an event after every step gives the per-step
    # times (median) without a host synchronisation inside the region
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]

    # ---- roofline pass: the same step, eager, only the dominant kernel and the first-layer DualQ-Conv bracketed
    summ = {}
    if timing and dominant and args.roofline_steps > 0:
        first = first_layer_label(survey)
        H.kernel_timer.only = {dominant} | ({first} if first else set())
        H.kernel_timer.active = True
        for _ in range(args.roofline_steps):
            eager_step()
        torch.cuda.synchronize()
        H.kernel_timer.active = False
        summ = H.kernel_timer.summary()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "train-step samples/sec (8ch x 128mel x 512T)", "value": round(batch * world * args.steps / elapsed, 3),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": n_warm, "ms_per_step": round(ms, 3),
            "ms_per_step_median": round(statistics.median(per_step), 3),
            "ms_per_step_max": round(max(per_step), 3), "slow_steps": [i for i, v in enumerate(per_step) if v > 1.5 * statistics.median(per_step)],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['name']} train step, F={args.freq} T=512 (U=L={w['U'] * (args.freq // 128)}, SURVEY F3), "
                                   f"batch {batch}/GPU, random-init weights", "global_batch": batch * world,
                       "parallelism": f"dp{world}", "step_mode": step_mode,
                       "deterministic": bool(H.deterministic())},
            "loss": round(final_loss, 6),
            # world size as torch.distributed sees it after init (1 = no process group) and the backend that carried it
            "rccl_ranks": (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1),
            "dist_backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
        }
        if summ:
            def rates(label, d, nsteps):
                tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
                gb = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                return dict(kernel=label, calls_per_step=d["calls"] / nsteps, ms_per_step=round(d["ms"] / nsteps, 4),
                            avg_us=round(d["ms"] / d["calls"] * 1e3, 2), tflops=round(tf, 2), gbs=round(gb, 1))

            def roofline(label):
                d = summ[label]
                r = rates(label, d, args.roofline_steps)
                # roof that bounds the kernel: compare time at each peak
                t_hbm = d["bytes"] / (PEAK_HBM_GBS * 1e9)
                t_mfma = d["flops"] / (PEAK_FP32_MFMA_TFLOPS * 1e12)
                if t_hbm >= t_mfma:
                    roof = dict(bound="hbm", achieved=r["gbs"], peak=PEAK_HBM_GBS, unit="GB/s",
                                frac=round(r["gbs"] / PEAK_HBM_GBS, 4))
                else:
                    roof = dict(bound="mfma", achieved=r["tflops"], peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                                frac=round(r["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4))
                # the committed counter passes are of the default workload (c3, batch 32, F = 128)
                traffic, src = pmc_traffic(label) if (args.workload == "c3" and batch == 32 and args.freq == 128) else (None, None)
                if label.startswith("hcq_"):
                    # the fast-product kernels EXECUTE half of the algorithmic flops the roofline is quoted in
                    # (8 real sub-products per Hamilton product instead of 16; 24 instead of 48 for the dual
                    # quaternion, csrc/hcq_conv.hip): `achieved` / `frac` follow SURVEY 8(d)'s algorithmic count and can
                    # exceed the MFMA peak; `executed_*` is what the matrix pipe actually does
                    roof.update(executed_flops_fraction=0.5, executed_achieved=round(r["tflops"] * 0.5, 2),
                                executed_frac=round(r["tflops"] * 0.5 / PEAK_FP32_MFMA_TFLOPS, 4))
                roof.update(kernel=label, avg_launch_us=r["avg_us"], launches_per_step=r["calls_per_step"],
                            traffic=traffic, traffic_source=src,
                            measured_over=f"{args.roofline_steps} instrumented eager steps after the timed region "
                                          f"(HIP events on the launch stream)")
                return roof, d
            if dominant in summ:
                out["roofline"], _ = roofline(dominant)
            first = first_layer_label(summ)
            if first and first != dominant:
                # the north-star's own target shape (first-layer DualQ-Conv forward), priced against HBM whatever the
                # formula above says: AI 26 puts it next to the ridge (19.7 flop/B)
                r, d = roofline(first)
                gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                r.update(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                         frac=round(gbs / PEAK_HBM_GBS, 4),
                         tflops=round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2))
                out["roofline_hbm"] = r
            # the other conv kernels: measured during warm-up steps 1..W-1 (every launch bracketed there)
            per = [rates(k, v, max(n_warm - 1, 1)) for k, v in survey.items()]
            per.sort(key=lambda r: -r["ms_per_step"])
            out["conv_kernels"] = per[:8]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, args.freq)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
