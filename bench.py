#!/usr/bin/env python3
"""bench.py — landmark-frames/sec of the Ishara CTC training step on MI355X.

A "step" is one full Keras train_step of the hot path on one synthetic batch: forward
(training=True, dropout on) + CTC loss + backward + [RCCL gradient all-reduce] +
Lookahead(RAdam) update, all in the HIP library.  Workload = BASELINE.json configs[1]:
get_model(dim=256, 2 squeeze + 2 conformer blocks, kernel_sizes=[11,5,3]) on B=256 clips per
GPU, T=384 frames, F=224 features, bf16 storage / fp32 accumulate, inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Rank 0 prints ONE JSON line (contract in the task brief) with two extra objects:
  roofline      the dominant kernel family of the step, timed live with HIP events on the
                launch stream (ishara_profile_*): algorithmic bytes per launch / avg duration
  cpu_baseline  the CPU oracle (oracle/ishara_oracle.py, a port — TensorFlow is not installed)
                timed on this host's cores on a bounded sample (B=8 clips of the same T,F,d)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA

MODEL_KW = dict(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, kernel_sizes=[11, 5, 3],
                num_conv_per_block=3, dropout_rate=0.2, num_heads=8, expansion_factor=2,
                transformer_kernel_size=15, input_shape=(384, 224))


def cpu_baseline(batch: int, steps: int = 12):      # ~11 s of CPU work on the 16-core GPU-box share
    """Full train step of the CPU oracle (fp32, torch-CPU, all host cores) on `batch` clips."""
    import numpy as np
    import torch
    from oracle import ishara_oracle as O
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))       # the GPU box gives one GPU a 16-core share
    cfg = O.Config(**{**MODEL_KW, "kernel_sizes": tuple(MODEL_KW["kernel_sizes"])})
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU (weak scaling)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from ishara_amd import get_model, parallel
    t_start = time.perf_counter()

    rank, world, local = parallel.init_from_env(args.backend)
    if args.share_gpu:
        local = 0
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    B, T, F = args.batch, MODEL_KW["input_shape"][0], MODEL_KW["input_shape"][1]
    model = get_model(**MODEL_KW, dtype=args.dtype, max_batch=B, device=dev, seed=0)   # identical replicas: shared seed
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
    dt = parallel.reduce_max(time.perf_counter() - t0, device=dev)
    loss_v = float(loss.item())

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
    roof.update(launches_per_step=rec["launches"], avg_launch_ms=avg_ms, algorithmic_bytes_per_launch=by, flops_per_launch=fl)
    # HBM traffic of that kernel per launch: from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same
    # command (tools/traffic.py applies the gfx950 corrections); PMC counters cannot be read live from inside the process.
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))["kernels"].get(fam)
        if tr and args.batch == 256 and args.dtype == "bf16":
            roof["traffic"] = tr["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    out = {
        "metric": "landmark-frames/sec training (B=256,T=384,d=256)", "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "configs[1]: get_model(dim=256, 2 squeeze + 2 conformer blocks, kernel_sizes=[11,5,3]) CTC train step "
                               "(fwd+CTC+bwd+RAdam/Lookahead, dropout on)", "batch_per_gpu": B, "global_batch": B * world,
                   "frames": T, "features": F, "params": model.n_total, "parallelism": f"dp{world}", "loss": loss_v},
        "roofline": roof,
        "kernels_ms": {k: round(v["ms"], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(8)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
