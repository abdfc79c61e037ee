#!/usr/bin/env python3
"""bench.py — landmark-frames/sec of the Ishara CTC training step on MI355X.

A "step" is one full Keras train_step of the hot path on one synthetic batch: forward
(training=True, dropout on) + CTC loss + backward + [RCCL gradient all-reduce] +
Lookahead(RAdam) update, all in the HIP library.  Default workload = BASELINE.json configs[1]:
get_model(dim=256, 2 squeeze + 2 conformer blocks, kernel_sizes=[11,5,3]) on B=256 clips per
GPU, T=384 frames, F=224 features, bf16 storage / fp32 accumulate, inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W [--config 2|4|5]

N>1: run under `python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU over RCCL); when started WITHOUT
that launcher, bench.py starts it itself as a child process and relays rank 0's JSON line and the exit code.  A mismatch
between --gpus and the ranks torch.distributed sees is an error, never a silent 1-GPU run.

--config 4: BASELINE configs[3] (d512, 6+6 blocks, 8 heads, T512) training throughput, B=512 per GPU.
--config 5: BASELINE configs[4] (inference, B=1, T=384, fp16 storage, hipGraph replay): metric = latency per clip.

Rank 0 prints ONE JSON line (contract in the task brief) with two extra objects:
  roofline      the dominant kernel family of the step, timed live with HIP events on the
                launch stream (ishara_profile_*): algorithmic bytes per launch / avg duration
  cpu_baseline  the CPU oracle (oracle/ishara_oracle.py, a port — TensorFlow is not installed)
                timed on this host's cores on a bounded sample (B=8 clips of the same T,F,d)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA

CONFIGS = {
    2: dict(kw=dict(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, kernel_sizes=[11, 5, 3], num_conv_per_block=3,
                    dropout_rate=0.2, num_heads=8, expansion_factor=2, transformer_kernel_size=15, input_shape=(384, 224)),
            batch=256, metric="landmark-frames/sec training (B=256,T=384,d=256)",
            workload="configs[1]: get_model(dim=256, 2 squeeze + 2 conformer blocks, kernel_sizes=[11,5,3]) CTC train step "
                     "(fwd+CTC+bwd+RAdam/Lookahead, dropout on)",
            train_bytes_per_frame=373e3, train_flops_per_frame=49.7e6),
    4: dict(kw=dict(dim=512, num_conv_squeeze_blocks=6, num_conv_conform_blocks=6, kernel_sizes=[11, 5, 3], num_conv_per_block=3,
                    dropout_rate=0.2, num_heads=8, expansion_factor=2, transformer_kernel_size=15, input_shape=(512, 224)),
            batch=512, metric="landmark-frames/sec training (B=512,T=512,d=512, 6+6 blocks)",      # B: SURVEY §8(d) "as large as fits" — 512 (workspace ~160 GiB of 288); 768 also fits (+1.3 % frames/s), 64 / 128 / 256 give 0.85 / 0.90 / 0.97 of its rate
            workload="configs[3]: get_model(dim=512, 6 squeeze + 6 conformer blocks, 8 heads) T=512 CTC train step "
                     "(fwd+CTC+bwd+RAdam/Lookahead, dropout on)",
            train_bytes_per_frame=2.18e6, train_flops_per_frame=563.2e6),
}


def cpu_baseline(model_kw, batch: int, steps: int = 12):      # ~11 s of CPU work on the 16-core GPU-box share
    """Full train step of the CPU oracle (fp32, torch-CPU, all host cores) on `batch` clips."""
    import numpy as np
    import torch
    from oracle import ishara_oracle as O
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))       # the GPU box gives one GPU a 16-core share
    cfg = O.Config(**{**model_kw, "kernel_sizes": tuple(model_kw["kernel_sizes"])})
    W = O.init_params(cfg, 0)
    names = [n for n, _, _, t in O.param_specs(cfg) if t]
    theta = np.concatenate([W[n].reshape(-1) for n in names])
    st = O.optimizer_init(theta)
    x, y = O.synthetic_batch(cfg, batch, 1)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        _, _, grads, stats = O.loss_and_grads(W, x, y, cfg, training=True, seed=2 + i)
        g = np.concatenate([grads[n].reshape(-1) for n in names])
        theta = O.optimizer_step(theta, g, st, lr=1e-3)
        off = 0
        for n in names:
            W[n] = theta[off:off + W[n].size].reshape(W[n].shape); off += W[n].size
        W.update(stats)
        times.append(time.perf_counter() - t0)
    med = sorted(times[1:])[len(times[1:]) // 2]
    return dict(value=batch * cfg.T / med, unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle fp32 train step, B={batch} clips x T={cfg.T} (same model/T/F as the GPU run), "
                       f"1 warm-up + {steps} timed steps, median {med:.2f} s/step")


def committed_traffic(family: str, tag: str):
    """HBM bytes per launch of `family` from the committed rocprofv3 --pmc passes (tools/traffic.py), but ONLY when that table
    was measured on the sources this .so was built from (source_hash stamp); PMC counters cannot be read live in-process."""
    from ishara_amd.build import source_hash
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{tag}.json")), key=lambda f: int(os.path.basename(f)[1:].split("_")[0]))
    path = cands[-1] if cands else os.path.join(ROOT, "profiles", f"traffic_{tag}.json")      # the latest round's table
    try:
        tab = json.load(open(path))
    except (OSError, ValueError):
        return None, f"no committed PMC table (profiles/r*_traffic_{tag}.json)"
    if tab.get("source_hash") != source_hash():
        return None, f"{os.path.basename(path)} was measured on another build (stamp {tab.get('source_hash')} != {source_hash()}): not quoted"
    kern = tab.get("kernels", {})
    rec = kern.get(family)
    if not rec:                                   # the HIP-event profiler keys a kernel by its name without defaulted template arguments
        cands = [v for k, v in kern.items() if k.startswith(family.rstrip(">"))]
        rec = max(cands, key=lambda v: v["launches"]) if cands else None
    if not rec:
        return None, f"{family} not in {os.path.basename(path)}"
    return rec["hbm_bytes_per_launch"], f"committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE pass of this build ({os.path.basename(path)}, stamp {tab['source_hash']}); not live"


def relaunch_distributed(args) -> int:
    """`python bench.py --gpus N` without the launcher: start torch.distributed.run as a child (nothing here has touched the GPU
    yet) and relay its output and exit code."""
    port = 29400 + os.getpid() % 500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def check_world(gpus: int, world: int) -> None:
    if world != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but torch.distributed sees WORLD_SIZE={world}: refusing to report a {world}-rank run as {gpus} GPUs")


def roofline_of(prof, cfg, frames_per_s):
    fam, rec = max(prof.items(), key=lambda kv: kv[1]["ms"])
    avg_ms = rec["ms"] / rec["launches"]
    by, fl = rec["bytes"] / rec["launches"], rec["flops"] / rec["launches"]
    ai = fl / by if by else 0.0
    if ai > MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        ach = fl / (avg_ms * 1e-3) / 1e12
        roof = dict(bound="mfma", kernel=fam, achieved=ach, peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / MFMA_PEAK_TFLOPS, traffic=None)
    else:
        ach = by / (avg_ms * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=fam, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None)
    roof.update(launches_per_step=rec["launches"], avg_launch_ms=avg_ms, algorithmic_bytes_per_launch=by, flops_per_launch=fl,
                # whole step against both roofs (SURVEY §8d algorithmic bytes / flops per frame x frames/s)
                step_hbm_frac=frames_per_s * cfg["train_bytes_per_frame"] / (HBM_PEAK_GBS * 1e9),
                step_mfma_frac=frames_per_s * cfg["train_flops_per_frame"] / (MFMA_PEAK_TFLOPS * 1e12))
    return fam, roof


def run_training(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from ishara_amd import get_model, parallel
    t_start = time.perf_counter()
    cfg = CONFIGS[args.config]
    rank, world, local = parallel.init_from_env(args.backend)
    check_world(args.gpus, world)
    if args.share_gpu:
        local = 0
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    B = args.batch or cfg["batch"]
    T, F = cfg["kw"]["input_shape"]
    model = get_model(**cfg["kw"], dtype=args.dtype, max_batch=B, device=dev, seed=0)   # identical replicas: shared seed
    model.optimizer.learning_rate = 1e-3
    g = np.random.default_rng(1 + rank)                                               # per-rank data shard
    x = torch.from_numpy(g.standard_normal((B, T, F)).astype(np.float32)).to(dev)
    y = np.full((B, 64), 59, np.int64)
    for b in range(B):
        n = int(g.integers(8, 32))
        y[b, :n] = g.integers(0, 59, n)
    y = torch.from_numpy(y).to(dev)

    def log(msg):
        if args.verbose and rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)
    log("model + data ready")
    for i in range(args.warmup):
        model.train_on_batch(x, y)
        if args.verbose: torch.cuda.synchronize(); log(f"warmup step {i} done")
    parallel.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = model.train_on_batch(x, y)
    torch.cuda.synchronize(); parallel.barrier()
    dt_local = time.perf_counter() - t0
    dt = parallel.reduce_max(dt_local, device=dev)
    loss_v = float(loss.item())
    # N > 1 diagnostics (outside the timed region, every rank takes part): each rank's own step time, the flat gradient all-reduce alone,
    # and the step without the exchange -> how much of the all-reduce the step exposes
    diag = None
    if world > 1:
        per_rank = torch.zeros(world, dtype=torch.float64, device=dev)
        per_rank[rank] = dt_local / args.steps * 1e3
        dist.all_reduce(per_rank)
        k = max(3, min(10, args.steps))
        flat = model.grads[:model.n_train]
        parallel.barrier(); torch.cuda.synchronize(); ta = time.perf_counter()
        for _ in range(k):
            parallel.allreduce_sum_(flat)
        torch.cuda.synchronize(); ar_ms = parallel.reduce_max(time.perf_counter() - ta, device=dev) / k * 1e3
        parallel.barrier(); torch.cuda.synchronize(); tb = time.perf_counter()
        for _ in range(k):
            model.train_on_batch(x, y, allreduce=False)
        torch.cuda.synchronize(); noar_ms = parallel.reduce_max(time.perf_counter() - tb, device=dev) / k * 1e3
        diag = {"per_rank_ms_per_step": [round(float(v), 4) for v in per_rank.cpu()], "allreduce_alone_ms": ar_ms,
                "allreduce_bytes": int(flat.numel()) * 4, "ms_per_step_without_allreduce": noar_ms,
                "exposed_allreduce_ms": dt / args.steps * 1e3 - noar_ms, "diagnostic_steps": k}

    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step")
    # one more step with every launch bracketed by HIP events; EVERY rank runs it (the step contains the gradient
    # all-reduce: a rank that skipped it would leave the others blocked in the collective), rank 0 reports
    prof = model.profile_step(x, y)
    parallel.barrier()
    log("profile: " + json.dumps({k: round(v["ms"], 3) for k, v in (prof or {}).items()}))
    if rank != 0:
        return
    ms = dt / args.steps * 1e3
    value = world * B * T * args.steps / dt
    fam, roof = roofline_of(prof, cfg, value / world)
    # the profiled step brackets every launch with events and therefore runs the weight-gradient slab sums as launches of their own
    # (`reduce_slabs(wgrad)`); in the timed steps they ride as extra workgroups of the next weight-gradient GEMM: kernels_ms is the
    # non-deferred launch sequence and does not sum to ms_per_step
    roof["profiled_path"] = "non-deferred slab sums (timed steps: deferred riders)"
    if B == cfg["batch"] and args.dtype == "bf16":
        roof["traffic"], roof["traffic_source"] = committed_traffic(fam, f"cfg{args.config}")
    else:
        roof["traffic_source"] = "not the profiled batch/dtype"
    out = {
        "metric": cfg["metric"], "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": cfg["workload"], "batch_per_gpu": B, "global_batch": B * world,
                   "frames": T, "features": F, "params": model.n_total, "parallelism": f"dp{world}", "loss": loss_v,
                   "ranks_seen": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else None),
                   "allreduce": ("bucketed+overlapped" if parallel.overlap_enabled() else "flat") if world > 1 else None,
                   "multi_gpu_diagnostics": diag},
        "roofline": roof,
        "kernels_ms": {k: round(v["ms"], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg["kw"], 8 if args.config == 2 else 2, 12 if args.config == 2 else 3)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5], help="BASELINE.json config number (1-based): 2 = configs[1] (the metric's), 4 = d512 6+6 T512, 5 = B=1 inference latency")
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (weak scaling); 0 = the config's")
    ap.add_argument("--dtype", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_distributed(args))
    if args.config == 5:
        if args.gpus != 1:
            raise SystemExit("bench.py: --config 5 (B=1 inference latency) does not shard: replicas only, run it with --gpus 1")
        from tools.bench_infer import run_inference_bench
        args.dtype = args.dtype or "f16"
        print(json.dumps(run_inference_bench(args)))
        return
    args.dtype = args.dtype or "bf16"
    run_training(args)


if __name__ == "__main__":
    main()
