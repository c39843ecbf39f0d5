#!/usr/bin/env python3
"""Benchmark of the hot path: Gauss-ILRMA (IP) iterations/s on synthetic spectrograms.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one GaussILRMA.update_once() (source model + spatial model + power normalisation, loss recording
off) over this rank's batch of utterances, inputs resident in HBM.  N=1 workload = BASELINE.json config 4
(M=4, F=1025, T=4096, K=4, one utterance); N>1 = the same per-GPU workload on every rank (independent utterances,
no data-path collective: "weak" scaling); value = utterance-iterations/s over all ranks.

Rank 0 prints ONE JSON line with `roofline` (covariance-accumulate kernel, HIP events on the launch stream) and,
at N=1, `cpu_baseline` (the NumPy oracle on the same workload, timed on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=500)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--dtype", default="float64", choices=["float64", "float32"],
                   help="storage/compute type of the kernels; float64 = the reference's complex128 path")
    p.add_argument("--utterances-per-gpu", type=int, default=1)
    p.add_argument("--channels", type=int, default=4)
    p.add_argument("--bins", type=int, default=1025)
    p.add_argument("--frames", type=int, default=4096)
    p.add_argument("--basis", type=int, default=4)
    p.add_argument("--power-statistic", default="covariance", choices=["covariance", "direct"])
    p.add_argument("--kernel-reps", type=int, default=50, help="launches of the covariance kernel for the roofline leg")
    p.add_argument("--cpu-iters", type=int, default=4, help="timed oracle iterations for cpu_baseline (0 = skip)")
    p.add_argument("--with-loss", action="store_true", help="also report it/s with recordable_loss=True")
    return p.parse_args()


def synth_mixture(torch, dev, B, M, F, T, seed):
    """Seeded synthetic convolutive mixture, generated on the device: sparse-envelope complex Gaussian sources
    mixed per bin by a random M x M matrix (so bins are correlated and the loss actually moves)."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    S = torch.randn((B, M, F, T), dtype=torch.float64, device=dev, generator=gen) + \
        1j * torch.randn((B, M, F, T), dtype=torch.float64, device=dev, generator=gen)
    env = torch.rand((B, M, 1, T), dtype=torch.float64, device=dev, generator=gen) ** 2
    A = torch.randn((B, F, M, M), dtype=torch.float64, device=dev, generator=gen) + \
        1j * torch.randn((B, F, M, M), dtype=torch.float64, device=dev, generator=gen)
    X = torch.einsum("bfmn,bnft->bmft", A, S * env).contiguous()
    del S, A
    return X


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from audio_source_separation_amd import distributed as D
    rank, world, local_rank = D.init_from_env(backend="nccl" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    n_gpus = world
    dev = torch.device("cuda", local_rank if world > 1 else torch.cuda.current_device())

    from audio_source_separation_amd.bss.ilrma import GaussILRMA

    B, M, F, T, K = args.utterances_per_gpu, args.channels, args.bins, args.frames, args.basis
    X = synth_mixture(torch, dev, B, M, F, T, seed=1000 + rank)
    cplx = torch.complex128 if args.dtype == "float64" else torch.complex64
    Xrun = X.to(cplx).contiguous()

    def make_model(record_loss):
        np.random.seed(111 + rank)
        m = GaussILRMA(n_basis=K, recordable_loss=record_loss, dtype=args.dtype, device=dev,
                       power_statistic=args.power_statistic)
        m.input = Xrun
        m._reset()
        return m

    def barrier():
        D.barrier(dev)

    def timed_steps(model, steps, warmup, with_loss=False):
        for _ in range(warmup):
            model.update_once()
            if with_loss:
                model._record_loss()  # exactly what GaussILRMA.__call__ does per iteration (value stays in HBM)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.update_once()
            if with_loss:
                model._record_loss()
        barrier()
        return D.max_over_ranks(time.perf_counter() - t0, device=dev)

    model = make_model(False)
    elapsed = timed_steps(model, args.steps, args.warmup)
    model._check_status()
    total_units = n_gpus * B * args.steps  # utterance-iterations
    value = total_units / elapsed

    extra = {}
    if args.with_loss:
        ml = make_model(True)
        dt = timed_steps(ml, args.steps, args.warmup, with_loss=True)
        extra["value_with_loss"] = n_gpus * B * args.steps / dt

    # ---------------- roofline leg: the covariance-accumulate kernel alone, HIP events on the launch stream
    roofline = None
    if rank == 0:
        eng = model._engine
        c = 16 if args.dtype == "float64" else 8
        r = c // 2
        # algorithmic bytes per launch (weights rebuilt in-kernel from Tb,V), per utterance x B
        bytes_per_launch = B * (M * F * T * c + (M * F * K + M * K * T) * r + M * F * M * M * c)
        kernel_name = "cov_stream_kernel"
        if K > 4:
            rpi = 2 if r == 8 else 4                      # rows per LDS-direct instruction (csrc/assx_cov_wide.hpp)
            lds = 2 * ((M * K + rpi - 1) // rpi * rpi) * 64 * r + 8 * M * K * r
            if lds <= 144 * 1024 and os.environ.get("ASSX_COV_WIDE", "1") != "0":
                # activation tile shared by 8 bins through LDS: still the "weights rebuilt in-kernel" contract
                kernel_name = "cov_wide_kernel (+ cov_wide_finalize_kernel)"
            else:
                # the source variance is materialised first (write N.F.T reals), then read back as (N,F,T) weights --
                # the "weights materialised" contract of SURVEY.md 8d plus the map's own write
                bytes_per_launch += B * 2 * M * F * T * r
                kernel_name = "source_variance_map_kernel + cov_stream_kernel (N,F,T weights)"
        for _ in range(5):
            eng.ilrma_cov_partials(model._X, model._Td, model._Vd)
        stream = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.kernel_reps):
            eng.ilrma_cov_partials(model._X, model._Td, model._Vd)
        e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / args.kernel_reps
        achieved = bytes_per_launch / (ms * 1e-3) / 1e9
        # HBM traffic per launch from the committed PMC passes (tools/pmc_traffic.py; same kernel, same shape)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "cov_traffic.json")))[args.dtype]
            if (M, F, T, K) == (4, 1025, 4096, 4):
                traffic = int(round(tj["traffic_bytes"] * B))
        except Exception:
            pass
        roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "kernel_ms": round(ms, 5), "algorithmic_bytes": bytes_per_launch}

    # ---------------- CPU baseline: the NumPy oracle on the same workload (rank 0, N=1 only)
    cpu_baseline = None
    if rank == 0 and n_gpus == 1 and args.cpu_iters > 0:
        from oracle import oracle_np as orc  # reported baseline only; never on the product path
        try:
            from threadpoolctl import threadpool_info
            cores = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
        except Exception:
            cores = os.cpu_count()
        Xh = X[0].cpu().numpy()
        np.random.seed(111)
        Tb, V = np.random.rand(M, F, K), np.random.rand(M, K, T)
        W = np.tile(np.eye(M, dtype=np.complex128), (F, 1, 1))
        W, Tb, V, _ = orc.ilrma_update_once(Xh, W, Tb, V)  # warm-up (page faults, thread pools)
        t0 = time.perf_counter()
        for _ in range(args.cpu_iters):
            W, Tb, V, _ = orc.ilrma_update_once(Xh, W, Tb, V)
        dt = time.perf_counter() - t0
        cpu_baseline = {"value": round(args.cpu_iters / dt, 4), "unit": "iterations/s", "cores": int(cores),
                        "kind": "port",
                        "sample": "%d full update_once() iterations of the NumPy oracle (streaming covariance) on one "
                                  "M=%d F=%d T=%d K=%d complex128 utterance, after 1 warm-up" % (args.cpu_iters, M, F, T, K)}

    if rank == 0:
        out = {
            "metric": "ILRMA iterations/sec (4ch, F=1025, T=4096)",
            "value": round(value, 2),
            "unit": "iterations/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.dtype == "float64" else "f32",
            "data": "synthetic",
            "config": {"workload": "gauss_ilrma_ip update_once, M=%d F=%d T=%d K=%d, normalize=power, loss off" % (M, F, T, K),
                       "utterances_per_gpu": B, "power_statistic": args.power_statistic,
                       "parallelism": "utterance-sharded x%d, no data-path collective" % n_gpus},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        out.update(extra)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
