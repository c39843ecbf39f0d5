#!/usr/bin/env python3
"""Benchmark of the hot path: Gauss-ILRMA (IP) iterations/s on synthetic spectrograms.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (re-executes this file under
torch.distributed.run on 127.0.0.1); under an external launcher WORLD_SIZE must equal N.  One process per GPU, RCCL.

One "step" = one GaussILRMA.update_once() (source model + spatial model + power normalisation, loss recording
off) over this rank's batch of utterances, inputs resident in HBM.  N=1 workload = BASELINE.json config 4
(M=4, F=1025, T=4096, K=4, one utterance); N>1 = the same per-GPU workload on every rank (independent utterances,
no data-path collective: "weak" scaling); value = utterance-iterations/s over all ranks.

Rank 0 prints ONE JSON line with
  roofline      covariance-accumulate kernel alone, HIP events on the launch stream, one utterance (X = 268.7 MB,
                within 0.1 % of the 256 MiB Infinity Cache) -- and `roofline_b8`, the same kernel on 8 utterances in
                one launch (2.15 GB: nothing of X survives in the cache), the figure to quote for sustained HBM rate;
  cpu_baseline  (N=1) the NumPy oracle on the same workload timed on this host: streaming form (`value`) and the
                reference's own materialising XX/R form on a quarter of the frames (`reference_form`);
  config5       (N>1, or --config5) 64 seeded utterances block-partitioned over the ranks, one batched launch
                sequence per rank, the reference's default 100 iterations + final projection back:
                `value` without the edges, `value_incl_edges` with scatter X / gather Y over RCCL.
"""
import argparse
import gc
import json
import os
import platform
import socket
import subprocess
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
    p.add_argument("--roofline-b8", type=int, default=8, help="utterances of the beyond-cache roofline leg (0 = skip)")
    p.add_argument("--cpu-iters", type=int, default=4, help="timed oracle iterations for cpu_baseline (0 = skip)")
    p.add_argument("--cpu-reference-form", default="quarter", choices=["quarter", "full", "off"],
                   help="reference-form (materialising XX/R) CPU leg: on T/4 frames (default), the full shape, or not")
    p.add_argument("--with-loss", type=int, default=1, nargs="?", const=1,
                   help="1 (default): also report it/s with recordable_loss=True, the reference's default (value_with_loss; "
                        "100-step side leg unless --steps is smaller)")
    p.add_argument("--with-b8", type=int, default=1,
                   help="1 (default): also time 8 utterances per launch -- BASELINE config 5's per-GPU regime, 2.15 GB of X, "
                        "nothing of it survives in the Infinity Cache from pass to pass (value_b8, utterance-iterations/s; < 1 s)")
    p.add_argument("--with-f32", type=int, default=1,
                   help="1 (default): also time the float32 storage mode on the same workload (extra line value_f32; < 1 s)")
    p.add_argument("--with-default-basis", type=int, default=1,
                   help="1 (default): also time the reference's default n_basis = 10 (ilrma.py:183), loss off and on "
                        "(value_k10, value_k10_with_loss; < 1 s)")
    p.add_argument("--with-other-configs", type=int, default=1,
                   help="1 (default): also time BASELINE configs 1-3 on rank 0 (side lines other_configs: EUC-NMF 513x256 n_basis 8 "
                        "and AuxLaplaceIVA M=2 1025x2048 as one library call each, IS-NMF 1025x4096 n_basis 32 per update; < 1 s)")
    p.add_argument("--prewarm-ms", type=float, default=150.0,
                   help="untimed load before the W warm-up steps: the clocks take ~100 ms of work to settle "
                        "(20 timed steps after 5 / 50 / 500 warm-up steps: 0.2025 / 0.1986 / 0.1945 ms per step)")
    p.add_argument("--measure-traffic", type=int, default=1,
                   help="1 (default; rank 0 at N = 1, after every timed leg): HBM traffic of the covariance kernel from two "
                        "rocprofv3 --pmc passes of tools/microbench.py run as child processes (FETCH_SIZE, WRITE_SIZE; "
                        "tools/pmc_traffic.py: measure_case) -> roofline.traffic of THIS run; falls back to "
                        "profiles/cov_traffic.json when rocprofv3 is missing or fails (traffic_source says which)")
    p.add_argument("--config5", default="auto", choices=["auto", "on", "off"],
                   help="config-5 leg (64 utterances sharded over the ranks); auto = only when N > 1")
    p.add_argument("--config5-utterances", type=int, default=64)
    p.add_argument("--config5-iterations", type=int, default=100)
    p.add_argument("--config5-timeout", type=int, default=420, help="seconds before the config-5 leg is abandoned")
    # test scaffolding (tests/test_gpu_multi.py): N ranks on ONE GPU with gloo-staged edges, so that the whole N > 1
    # flow of this file runs on a 1-GPU box.  Never used for a reported number.
    p.add_argument("--comm-backend", default="nccl", choices=["nccl", "gloo"])
    p.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (only with --comm-backend gloo)")
    return p.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks (one per GPU) and relay their output."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not (args.share_gpu and args.comm_backend == "gloo" and have >= 1):
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible; refusing to run fewer ranks "
                         "than asked for.\n" % (args.gpus, have))
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


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


def cov_contract_bytes(B, M, F, T, K, r, lds_ok=True):
    """Algorithmic bytes of ONE covariance-accumulate launch (weights rebuilt in-kernel from Tb, V)."""
    c = 2 * r
    nbytes = B * (M * F * T * c + (M * F * K + M * K * T) * r + M * F * M * M * c)
    name = "cov_stream_kernel"
    if K > 4:
        rpi = 2 if r == 8 else 4                      # rows per LDS-direct instruction (csrc/assx_cov_wide.hpp)
        lds = 2 * ((M * K + rpi - 1) // rpi * rpi) * 64 * r + 8 * M * K * r
        if lds <= 144 * 1024:
            # still "weights rebuilt in-kernel"; n_basis <= 16 runs the matrix-core variance form since round 3
            # (csrc/assx_cov_mfma.hpp)
            mfma = K <= 16
            name = ("cov_mfma_kernel" if mfma else "cov_wide_kernel") + " (+ cov_wide_finalize_kernel)"
        else:
            # the source variance is materialised first (write N.F.T reals), then read back as (N,F,T) weights --
            # the "weights materialised" contract of SURVEY.md 8d plus the map's own write
            nbytes += B * 2 * M * F * T * r
            name = "source_variance_map_kernel + cov_stream_kernel (N,F,T weights)"
    return nbytes, name


def time_cov_kernel(torch, eng, X, Td, Vd, reps):
    """Mean duration (ms) of assx_ilrma_cov_partials = exactly one launch of the covariance kernel."""
    for _ in range(5):
        eng.ilrma_cov_partials(X, Td, Vd)
    stream = torch.cuda.current_stream(eng.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        eng.ilrma_cov_partials(X, Td, Vd)
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def host_description():
    model = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    info = {"cpu_model": model, "logical_cpus": os.cpu_count(), "numpy": np.__version__}
    try:
        from threadpoolctl import threadpool_info
        pools = threadpool_info()
        info["blas"] = ["%s %s (%s threads)" % (p.get("internal_api"), p.get("version"), p.get("num_threads"))
                        for p in pools if p.get("user_api") == "blas"]
        info["threads"] = max([p.get("num_threads", 1) for p in pools] + [1])
    except Exception:
        info["blas"] = []
        info["threads"] = os.cpu_count()
    return info


def cpu_baseline_leg(args, Xh, M, F, T, K):
    """The oracle (a NumPy port of the reference's update_once) on the GPU box's host cores; a reported baseline,
    not the optimisation target.  Streaming form on the full utterance; reference (materialising) form on T/4."""
    from oracle import oracle_np as orc  # reported baseline only; never on the product path
    host = host_description()
    np.random.seed(111)
    Tb, V = np.random.rand(M, F, K), np.random.rand(M, K, T)
    W = np.tile(np.eye(M, dtype=np.complex128), (F, 1, 1))
    W, Tb, V, _ = orc.ilrma_update_once(Xh, W, Tb, V)  # warm-up (page faults, thread pools)
    t0 = time.perf_counter()
    for _ in range(args.cpu_iters):
        W, Tb, V, _ = orc.ilrma_update_once(Xh, W, Tb, V)
    dt = time.perf_counter() - t0
    out = {"value": round(args.cpu_iters / dt, 4), "unit": "iterations/s", "cores": int(host["threads"]),
           "kind": "port",
           "sample": "%d full update_once() iterations of the NumPy oracle (streaming covariance) on one "
                     "M=%d F=%d T=%d K=%d complex128 utterance, after 1 warm-up" % (args.cpu_iters, M, F, T, K),
           "host": host}
    if args.cpu_reference_form != "off":
        Ts = T if args.cpu_reference_form == "full" else max(T // 4, 1)
        Xs = np.ascontiguousarray(Xh[:, :, :Ts])
        np.random.seed(111)
        Tb, V = np.random.rand(M, F, K), np.random.rand(M, K, Ts)
        W = np.tile(np.eye(M, dtype=np.complex128), (F, 1, 1))
        orc.ilrma_update_once_reference_form(Xs, W, Tb, V)  # warm-up: first-touch of the 1 GB temporaries, BLAS threads
        dts = []
        for _ in range(2):
            t0 = time.perf_counter()
            orc.ilrma_update_once_reference_form(Xs, W, Tb, V)
            dts.append(time.perf_counter() - t0)
        dt = min(dts)
        out["reference_form"] = {
            "value": round(Ts / T / dt, 4), "unit": "iterations/s (scaled to T=%d)" % T, "seconds_measured": round(dt, 2),
            "sample": "best of 2 update_once() after 1 warm-up in the reference's materialising form (XX (F,T,M,M), "
                      "XX/R (N,F,T,M,M) = %.2f GB, .mean) on M=%d F=%d T=%d of the same utterance; cost is linear in T, "
                      "value = (T_sample/T)/seconds" % (M * F * Ts * M * M * 16 / 1e9, M, F, Ts)}
    return out


def config5_leg(args, torch, D, dev, comm_dev, rank, world, M, F, T, K):
    """64 seeded utterances -> shard_range blocks -> one batched GaussILRMA call per rank -> gather."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    U, iters = args.config5_utterances, args.config5_iterations
    cplx = torch.complex128 if args.dtype == "float64" else torch.complex64
    x_all = None
    lo, hi = D.shard_range(U, world, rank)
    # Footprint in HBM.  Rank 0 holds the whole batch twice (x_all for the scatter, y_all from the gather) next to its own
    # block: config 5 in complex128 = 2 x 64 x 268.7 MB = 34.4 GB + a block of 8 (x_local 2.15 GB + y_local 2.15 GB + the
    # model's workspace, < 1 GB) -- 39 GB of 288; a peer holds only its block.  Checked before anything is allocated, so
    # that an oversized --config5-utterances fails with a message instead of an allocator error in the middle of a leg.
    csize = 16 if cplx == torch.complex128 else 8
    need = 2 * (hi - lo) * M * F * T * csize + (1 << 30)
    if rank == 0 and comm_dev == dev:
        need += 2 * U * M * F * T * csize
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    short = need > free_b + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    if D.max_over_ranks(1.0 if short else 0.0, device=dev) > 0:  # every rank leaves together: nobody waits in a scatter
        raise MemoryError("config-5 leg: some rank is short of HBM (rank %d needs %.1f GB -- x_all + y_all on the root, its "
                          "own block, workspace -- and has %.1f GB free)" % (rank, need / 1e9, free_b / 1e9))
    if rank == 0:
        x_all = torch.empty((U, M, F, T), dtype=cplx, device=dev)
        for u0 in range(0, U, 8):  # utterance u has seed u ("64 utterances of cfg4 with seeds 0..63")
            for u in range(u0, min(u0 + 8, U)):
                x_all[u] = synth_mixture(torch, dev, 1, M, F, T, seed=u)[0].to(cplx)

    def init_fn(model, lo, hi):
        st = [np.random.RandomState(111 + u) for u in range(lo, hi)]
        model.basis = np.stack([s.rand(M, F, K) for s in st])
        model.activation = np.stack([s.rand(M, K, T) for s in st])

    def factory():
        return GaussILRMA(n_basis=K, recordable_loss=False, dtype=args.dtype, device=dev)

    # warm-up of the compute path at this batch size (workspace growth, clocks), outside every timed region
    if hi > lo:
        warm = factory()
        init_fn(warm, lo, hi)
        gen = torch.Generator(device=dev).manual_seed(7)
        xw = torch.view_as_complex(torch.randn((hi - lo, M, F, T, 2), dtype=torch.float64 if cplx == torch.complex128
                                               else torch.float32, device=dev, generator=gen))
        warm(xw, iteration=2)
        del warm, xw
    if rank == 0 and comm_dev != dev:
        x_all = x_all.to(comm_dev)  # gloo-staged test mode: the edges travel through host memory
    D.warm_up_edges(comm_dev)
    D.barrier(dev)
    t0 = time.perf_counter()
    x_local = D.scatter_utterances(x_all, U, (M, F, T), cplx, comm_dev).to(dev)
    D.barrier(dev)
    t1 = time.perf_counter()
    model = factory()
    if hi > lo:
        init_fn(model, lo, hi)
        y_local = model(x_local, iteration=iters)
    else:
        y_local = torch.empty((0, M, F, T), dtype=cplx, device=dev)
    D.barrier(dev)
    t2 = time.perf_counter()
    y_all = D.gather_utterances(y_local.to(comm_dev), U)
    D.barrier(dev)
    t3 = time.perf_counter()
    compute = D.max_over_ranks(t2 - t1, device=dev)
    total = D.max_over_ranks(t3 - t0, device=dev)
    if rank != 0:
        return None
    ok = bool(torch.isfinite(torch.view_as_real(y_all)).all().item())
    return {"utterances": U, "iterations": iters, "utterances_per_gpu": D.shard_sizes(U, world),
            "value": round(U * iters / compute, 2), "value_incl_edges": round(U * iters / total, 2),
            "unit": "utterance-iterations/s", "seconds_compute": round(compute, 4),
            "seconds_scatter": round(t1 - t0, 4), "seconds_gather": round(t3 - t2, 4),
            "edge_bytes": 2 * U * M * F * T * (16 if args.dtype == "float64" else 8) * (world - 1) // world,
            "outputs_finite": ok,
            "note": "GaussILRMA()(X_block, iteration=%d) per rank incl. the final projection back; edges = grouped "
                    "RCCL send/recv root<->peers of X and Y" % iters}


def other_configs_leg(torch, device):
    """Side lines for BASELINE configs 1-3 (never `value`): synthetic inputs of the quoted shapes, float64, loss off.
    config 1 / 3 run as ONE library call (what `model(X, iteration=k)` does without callbacks), config 2 is timed per
    update with HIP events (20 updates after 3)."""
    import gc
    from audio_source_separation_amd import _lib
    from audio_source_separation_amd.ops import Engine
    eng = Engine("float64", device=device)
    g = torch.Generator(device=eng.dev).manual_seed(0)
    out = {}
    gc.collect()
    gc.disable()
    try:
        # config 2: IS-NMF, F = 1025, T = 4096, n_basis = 32 (12 F T K flop per update)
        F, T, K = 1025, 4096, 32
        X = torch.rand((1, F, T), dtype=torch.float64, device=eng.dev, generator=g) ** 2
        Tb = torch.rand((1, F, K), dtype=torch.float64, device=eng.dev, generator=g)
        V = torch.rand((1, K, T), dtype=torch.float64, device=eng.dev, generator=g)
        for _ in range(3):
            eng.nmf_update(_lib.NMF_IS_MM, X, Tb, V)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.nmf_update(_lib.NMF_IS_MM, X, Tb, V)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out["config2_isnmf_update_us"] = round(us, 2)
        out["config2_isnmf_tflops"] = round(12.0 * F * T * K / us / 1e6, 2)
        # config 1: EUC-NMF 513 x 256, n_basis 8, 2000 updates in one call
        F, T, K, n = 513, 256, 8, 2000
        X = torch.rand((1, F, T), dtype=torch.float64, device=eng.dev, generator=g) ** 2
        Tb = torch.rand((1, F, K), dtype=torch.float64, device=eng.dev, generator=g)
        V = torch.rand((1, K, T), dtype=torch.float64, device=eng.dev, generator=g)
        eng.nmf_iterate(50, _lib.NMF_EUC, X, Tb, V)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.nmf_iterate(n, _lib.NMF_EUC, X, Tb, V)
        torch.cuda.synchronize()
        out["config1_eucnmf_updates_per_s"] = round(n / (time.perf_counter() - t0), 1)
        # config 3: AuxLaplaceIVA-IP, M = 2, F = 1025, T = 2048, 1000 iterations in one call
        M, F, T, n = 2, 1025, 2048, 1000
        S = torch.randn((M, F, T), dtype=torch.float64, device=eng.dev, generator=g) + \
            1j * torch.randn((M, F, T), dtype=torch.float64, device=eng.dev, generator=g)
        A = torch.randn((F, M, M), dtype=torch.complex128, device=eng.dev, generator=g)
        X = torch.einsum("fmn,nft->mft", A, S).contiguous()[None]
        W = torch.eye(M, dtype=torch.complex128, device=eng.dev).repeat(1, F, 1, 1).contiguous()
        r = eng.empty((1, M, T))
        st = eng.new_status(1)
        eng.auxiva_iterate(20, _lib.IVA_LAPLACE, X, W, r, status=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.auxiva_iterate(n, _lib.IVA_LAPLACE, X, W, r, status=st)
        torch.cuda.synchronize()
        out["config3_auxlaplaceiva_it_per_s"] = round(n / (time.perf_counter() - t0), 1)
    except Exception as exc:  # a side line never takes the headline down
        out["error"] = repr(exc)
    finally:
        gc.enable()
    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        self_launch(args)
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher set WORLD_SIZE=%s; they must agree.\n"
                         % (args.gpus, env_world))
        sys.exit(2)

    import torch
    import torch.distributed as dist

    from audio_source_separation_amd import distributed as D
    if args.share_gpu and args.comm_backend != "gloo":
        sys.stderr.write("bench.py: --share-gpu needs --comm-backend gloo (RCCL cannot put two ranks on one GPU).\n")
        sys.exit(2)
    rank, world, local_rank = D.init_from_env(backend=args.comm_backend if args.gpus > 1 else None)
    n_gpus = world
    dev = torch.device("cuda", 0 if args.share_gpu else (local_rank if world > 1 else torch.cuda.current_device()))
    comm_dev = dev if (world == 1 or args.comm_backend == "nccl") else torch.device("cpu")
    comm = {"backend": (dist.get_backend() if dist.is_initialized() else None),
            "world_size": (dist.get_world_size() if dist.is_initialized() else 1)}

    from audio_source_separation_amd.bss.ilrma import GaussILRMA

    B, M, F, T, K = args.utterances_per_gpu, args.channels, args.bins, args.frames, args.basis
    X = synth_mixture(torch, dev, B, M, F, T, seed=1000 + rank)
    cplx = torch.complex128 if args.dtype == "float64" else torch.complex64
    Xrun = X.to(cplx).contiguous()

    def make_model(record_loss, dtype=None, n_basis=None, x=None):
        np.random.seed(111 + rank)
        dtype = dtype or args.dtype
        m = GaussILRMA(n_basis=n_basis or K, recordable_loss=record_loss, dtype=dtype, device=dev,
                       power_statistic=args.power_statistic)
        if x is not None:
            m.input = x
        else:
            m.input = Xrun if dtype == args.dtype else X.to(torch.complex64 if dtype == "float32" else torch.complex128).contiguous()
        m._reset()
        return m

    def barrier():
        D.barrier(dev)

    def prewarm(model):
        """Bring the GPU to its steady state with the same work, before the contract's W warm-up steps: chunks of 20 steps for
        at least --prewarm-ms, then until two consecutive chunks take the same time within 1 % (at most 8 x --prewarm-ms).
        The part takes ~100 ms of load to settle as a rule, but a process that starts on a box that has just been idle -- what
        the driver's single run is -- was seen to need more: five runs of the driver's command in a row read 0.1875, 0.1805,
        0.1814, 0.1790, 0.1788 ms per step with the fixed 150 ms (profiles/r06_bench_driver_cmd_fourth_box.json)."""
        if args.prewarm_ms <= 0:
            return
        t_start = time.perf_counter()
        t_min, t_max = t_start + args.prewarm_ms * 1e-3, t_start + 8 * args.prewarm_ms * 1e-3
        prev = None
        while True:
            t0 = time.perf_counter()
            for _ in range(20):
                model.update_once()
            torch.cuda.synchronize(dev)
            now = time.perf_counter()
            dt = now - t0
            if now >= t_max or (now >= t_min and prev is not None and abs(dt - prev) <= 0.01 * prev):
                break
            prev = dt
        prewarm_spent[0] = max(prewarm_spent[0], (time.perf_counter() - t_start) * 1e3)

    def timed_steps(model, steps, warmup, with_loss=False):
        # like timeit: no cyclic garbage collection inside the timed region (a model of an earlier leg is a reference
        # cycle -- model <-> loss list -- and a full collection in the middle of a leg costs tens of ms of host time:
        # round 4 saw a side leg turn host-bound, 374 us per step, for exactly that reason).  The collection happens
        # HERE, before any GPU work of the leg: rounds 4-5 ran it between the last warm-up step and t0, i.e. the part
        # idled for the length of a full collection right before a region of a few ms (round 5's review, weak #1).
        gc.collect()
        gc.disable()
        prewarm(model)
        for _ in range(warmup):
            model.update_once()
            if with_loss:
                model._record_loss()  # exactly what GaussILRMA.__call__ does per iteration (value stays in HBM)
        barrier()  # the contract's bracket: barrier + synchronize, nothing else between the warm-up and t0
        t0 = time.perf_counter()
        for _ in range(steps):
            model.update_once()
            if with_loss:
                model._record_loss()
        host_loop[0] = (time.perf_counter() - t0) / steps  # enqueueing alone (diagnostic: is a leg host-bound?)
        barrier()
        dt = D.max_over_ranks(time.perf_counter() - t0, device=dev)
        gc.enable()
        return dt

    host_loop = [0.0]
    prewarm_spent = [0.0]

    model = make_model(False)

    # ---------------- roofline legs first (rank 0): the covariance-accumulate kernel alone, HIP events on the launch
    # stream.  Running them before the timed steps also brings the clocks up, so a short --steps run is not low.
    roofline = roofline_b8 = None
    r = 8 if args.dtype == "float64" else 4
    if rank == 0:
        eng = model._engine
        nbytes, kernel_name = cov_contract_bytes(B, M, F, T, K, r)
        ms = time_cov_kernel(torch, eng, model._X, model._Td, model._Vd, args.kernel_reps)
        achieved = nbytes / (ms * 1e-3) / 1e9
        # HBM traffic per launch: NOT measured in this run -- read from the committed PMC passes of the same kernel
        # and shape (tools/pmc_traffic.py -> profiles/cov_traffic.json); null when there is no matching record
        traffic = traffic_source = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "cov_traffic.json")))
            rec = tj[args.dtype] if K <= 4 else tj.get("wide_k%d" % K, {}).get(args.dtype)
            if rec and (M, F, T) == (4, 1025, 4096):
                traffic = int(round(rec["traffic_bytes"] * B))
                traffic_source = "profiles/cov_traffic.json (separate rocprofv3 --pmc passes, %s)" % rec.get(
                    "collected", "round 1")
        except Exception:
            pass
        roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": round(ms, 5),
                    "algorithmic_bytes": nbytes, "utterances_per_launch": B}
        if args.roofline_b8 > 0 and args.roofline_b8 != B:
            B8 = args.roofline_b8
            X8 = torch.cat([Xrun[:1]] * B8, dim=0).contiguous() if B == 1 else Xrun[:1].repeat(B8, 1, 1, 1)
            T8 = model._Td[:1].repeat(B8, 1, 1, 1).contiguous()
            V8 = model._Vd[:1].repeat(B8, 1, 1, 1).contiguous()
            nb8, _ = cov_contract_bytes(B8, M, F, T, K, r)
            ms8 = time_cov_kernel(torch, eng, X8, T8, V8, max(args.kernel_reps // 4, 5))
            a8 = nb8 / (ms8 * 1e-3) / 1e9
            roofline_b8 = {"bound": "hbm", "kernel": kernel_name, "achieved": round(a8, 1), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(a8 / HBM_PEAK_GBS, 4), "traffic": None,
                           "kernel_ms": round(ms8, 5), "algorithmic_bytes": nb8, "utterances_per_launch": B8,
                           "note": "working set %.2f GB >> 256 MiB Infinity Cache" % (nb8 / 1e9)}
            del X8, T8, V8
            torch.cuda.empty_cache()

    elapsed = timed_steps(model, args.steps, args.warmup)
    model._check_status()
    total_units = n_gpus * B * args.steps  # utterance-iterations
    value = total_units / elapsed

    extra = {}
    # side lines of the same workload (never `value`): the loss recorded (the reference's default), the float32 storage
    # mode, the reference's default n_basis, 8 utterances per launch
    side_steps, side_warm = min(args.steps, 100), min(args.warmup, 10)
    if args.with_loss:
        ml = make_model(True)
        extra["value_with_loss"] = round(n_gpus * B * side_steps / timed_steps(ml, side_steps, side_warm, with_loss=True), 2)
        del ml
    if args.with_f32 and args.dtype == "float64":
        m32 = make_model(False, dtype="float32")
        extra["value_f32"] = round(n_gpus * B * side_steps / timed_steps(m32, side_steps, side_warm), 2)
        del m32
    if args.with_default_basis and K != 10:
        for key, rl in (("value_k10", False), ("value_k10_with_loss", True)):
            mk = make_model(rl, n_basis=10)
            extra[key] = round(n_gpus * B * side_steps / timed_steps(mk, side_steps, side_warm, with_loss=rl), 2)
            extra[key + "_host_us_per_step"] = round(host_loop[0] * 1e6, 1)
            del mk
    if args.with_b8 and B == 1 and (M, F, T) == (4, 1025, 4096):
        B8 = 8
        X8 = synth_mixture(torch, dev, B8, M, F, T, seed=2000 + rank).to(cplx).contiguous()
        m8 = make_model(False, x=X8)
        s8, w8 = min(args.steps, 40), min(args.warmup, 5)
        extra["value_b8"] = round(n_gpus * B8 * s8 / timed_steps(m8, s8, w8), 2)
        extra["value_b8_note"] = "utterance-iterations/s, %d utterances per launch (config 5's per-GPU batch), %d steps" % (B8, s8)
        del m8, X8
    torch.cuda.empty_cache()
    if args.with_other_configs and rank == 0 and args.dtype == "float64":
        extra["other_configs"] = other_configs_leg(torch, dev)

    # ---------------- HBM traffic of the roofline kernel, measured now (every timed leg is over: the child processes and
    # their counters cannot disturb a figure above)
    if (rank == 0 and n_gpus == 1 and args.measure_traffic and roofline is not None and (M, F, T) == (4, 1025, 4096)
            and K in (4, 10) and B == 1):
        try:
            import importlib.util
            import shutil
            import tempfile
            if shutil.which("rocprofv3") is None:
                raise RuntimeError("rocprofv3 not on PATH")
            spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
            pt = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(pt)
            with tempfile.TemporaryDirectory(dir="/tmp") as wd:
                rec = pt.measure_case("k4" if K <= 4 else "k10", args.dtype, wd)
            if rec:
                roofline["traffic"] = int(round(rec["traffic_bytes"]))
                roofline["traffic_source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two "
                                              "child processes of tools/microbench.py, %d / %d launches averaged), x1024, fetch x2 "
                                              "(MI355X_MICROARCH.md, HBM section)" % tuple(rec["launches_averaged"]))
                roofline["traffic_over_algorithmic"] = round(rec["traffic_bytes"] / roofline["algorithmic_bytes"], 4)
        except Exception as exc:  # the committed figure stays, labelled as such
            roofline["traffic_live_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])

    # ---------------- CPU baseline: the NumPy oracle on the same workload (rank 0, N=1 only)
    cpu_baseline = None
    if rank == 0 and n_gpus == 1 and args.cpu_iters > 0:
        cpu_baseline = cpu_baseline_leg(args, X[0].cpu().numpy(), M, F, T, K)

    out = None
    if rank == 0:
        out = {
            "metric": "ILRMA iterations/sec (4ch, F=1025, T=4096)",
            "value": round(value, 2),
            "unit": "iterations/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_ms": args.prewarm_ms,
            "prewarm_ms_spent_max": round(prewarm_spent[0], 1),
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.dtype == "float64" else "f32",
            "data": "synthetic",
            "config": {"workload": "gauss_ilrma_ip update_once, M=%d F=%d T=%d K=%d, normalize=power, loss off" % (M, F, T, K),
                       "utterances_per_gpu": B, "power_statistic": args.power_statistic,
                       "parallelism": "utterance-sharded x%d, no data-path collective" % n_gpus},
            "comm": dict(comm, utterances_per_rank=[B] * n_gpus),
            "roofline": roofline,
            "roofline_b8": roofline_b8,
            "cpu_baseline": cpu_baseline,
        }
        out.update(extra)

    # ---------------- config 5: sharded batch of utterances with and without the RCCL edges.  The headline
    # measurement above is complete at this point; a watchdog makes sure its line is printed even if a point-to-point
    # edge of this leg were to hang (rank 0 prints the line with config5.error, every rank leaves).
    if args.config5 == "on" or (args.config5 == "auto" and n_gpus > 1):
        import threading

        def give_up():
            if rank == 0:
                out["config5"] = {"error": "timeout after %d s (the headline measurement is unaffected)" % args.config5_timeout}
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(args.config5_timeout, give_up)
        dog.daemon = True
        dog.start()
        del model
        torch.cuda.empty_cache()
        try:
            config5 = config5_leg(args, torch, D, dev, comm_dev, rank, world, M, F, T, K)
        except Exception as exc:  # the headline line must survive a failure of this leg
            config5 = {"error": "%s: %s" % (type(exc).__name__, exc)}
        dog.cancel()
        if rank == 0:
            out["config5"] = config5
        if config5 is not None and "error" in config5:  # peers may be stuck in a collective: no farewell barrier
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)

    if rank == 0:
        print(json.dumps(out), flush=True)

    if world > 1:
        D.barrier(dev)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
