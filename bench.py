#!/usr/bin/env python3
"""Headline benchmark: POD build throughput (snapshot-DoF/s) on the BASELINE.json workload.

Workload (BASELINE.json configs[2], the shape the metric and the north-star target are quoted
on; 4.1 GB, fits one GPU): synthetic snapshot matrix N_h = 1e6 x n_s = 512 (float64, prescribed
singular-value decay 10^(-8 i/511)), r = 40 POD modes, orth(normalize=True, num=40).
One "step" = one full orth-equivalent on snapshots already resident in HBM:
    Gram (FP64 MFMA) -> [all-reduce] -> normalise -> n x n eigensolve -> back-projection.
With --gpus N the rows are sharded over the N ranks (strong scaling: the global matrix is the
same for N = 1, 2, 4, 8) and the Gram matrix is all-reduced over RCCL.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no launcher: the script starts its N ranks itself, before any GPU call)

A line is only ever printed with n_gpus == --gpus; every line carries `fallback_counters` (eigensolver hand-off
time-outs, contexts that left the one-XCD hand-off form, sets recomputed on the regular route, summed over the ranks).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant
kernel (the Gram MFMA kernel, timed with HIP events on its own stream inside the timed region)
and `cpu_baseline` (the NumPy/SciPy oracle -- the reference's exact library calls -- on a bounded
row sample, rank 0, N = 1 only).

The metric has a second half, "reduced timesteps/s" (BASELINE.json configs[4]: 1e4 timesteps x 32 parameter
points, r = 80; single GPU by the north star).  At N = 1 the same process therefore also runs, after the POD
region, the online sweep of that configuration twice - the direct path (V^T(A V) on the matrix cores every step,
N_h = 1e5) and the hyper-reduced path ((M)DEIM expansions, nothing of size N_h) - and reports both under
`secondary`, each with its own `roofline` / `cpu_baseline` (--no-secondary skips it).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

N_H, N_S, R_MODES = 1_000_000, 512, 40
CHUNK_ROWS = 125_000          # the global matrix is generated in 8 fixed chunks (seed = base + chunk)
SEED = 20260104 + 3
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (AMD spec) -- the guide lists no f64 row
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s


def spectrum(n):
    return 10.0 ** (-8.0 * np.arange(n) / (n - 1))


def mixing_matrix(n, device):
    """diag(sigma) V0^T with V0 a fixed orthogonal matrix (same on every rank)."""
    g = torch.Generator(device="cpu").manual_seed(SEED)
    V0, _ = torch.linalg.qr(torch.randn((n, n), dtype=torch.float64, generator=g))
    return (torch.from_numpy(spectrum(n))[:, None] * V0.T).to(device)


def make_chunk(c, rows, n, mix, device):
    """Rows [c*CHUNK, c*CHUNK + rows) of the global snapshot matrix: (Z / sqrt(N_H)) diag(sigma) V0^T."""
    g = torch.Generator(device=device).manual_seed(SEED + 1 + c)
    Z = torch.randn((rows, n), dtype=torch.float64, device=device, generator=g)
    return (Z @ mix) * (1.0 / np.sqrt(N_H))


def build_local_matrix(rank, world, n_h, n, device):
    mix = mixing_matrix(n, device)
    n_chunks = (n_h + CHUNK_ROWS - 1) // CHUNK_ROWS
    per_rank = (n_chunks + world - 1) // world
    parts = []
    for c in range(rank * per_rank, min(n_chunks, (rank + 1) * per_rank)):
        rows = min(CHUNK_ROWS, n_h - c * CHUNK_ROWS)
        parts.append(make_chunk(c, rows, n, mix, device))
    X = torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]
    return X.contiguous()


def cpu_baseline(X_host, r, device_result=None):
    """The oracle's orth (column norms + divide + dgesvd + energy + truncation, pod.py:31-57) on the host copy of the
    workload's own snapshot matrix (or its first rows), all host cores.  With ``device_result`` (Q on the device, s on
    the host, of one of the timed PODs of the SAME matrix) the dgesvd result is not thrown away: `parity` is the
    distance between the two at the stated size."""
    from oracle import romtime_oracle as oracle

    sample_rows, n = X_host.shape
    cores = min(16, len(os.sched_getaffinity(0)))  # a 1-GPU box gives this process a 16-core share
    from threadpoolctl import threadpool_limits

    with threadpool_limits(limits=cores):
        t0 = time.perf_counter()
        Q, s, energy = oracle.orth(X_host, num=r, normalize=True)
        dt = time.perf_counter() - t0
    out = dict(value=sample_rows * n / dt, unit="snapshot-DoF/s", cores=cores, kind="port",
               sample=f"oracle.orth (scipy dgesvd) on {'the whole' if sample_rows >= N_H else 'the first'} {sample_rows} x {n} rows of the workload, "
                      f"{dt:.2f} s, numpy {np.__version__}")
    if device_result is not None:
        Qd, sd = device_result["Q"].cpu().numpy(), np.asarray(device_result["s"])
        eps = np.finfo(float).eps
        bar = 2e-13 * s[0] + 8 * eps * s[0] ** 2 / np.maximum(s, 1e-300)      # the bar of tests/test_surface.py
        with threadpool_limits(limits=cores):
            resid = Qd - Q @ (Q.T @ Qd)                                        # sine of the largest principal angle
            col = np.minimum(np.linalg.norm(Qd - Q, axis=0), np.linalg.norm(Qd + Q, axis=0)) if Qd.shape == Q.shape else [np.inf]
            out["parity"] = dict(
                sigma_max_rel=float(np.abs(sd - s).max() / s[0]), sigma_within_test_bar=bool(np.all(np.abs(sd - s) <= bar)),
                subspace_dist=float(np.linalg.norm(resid, 2)), max_column_dist_up_to_sign=float(np.max(col)),
                kept_modes_equal=bool(Qd.shape[1] == Q.shape[1]), kept_modes=int(Q.shape[1]),
                energy_max_abs=float(np.abs(np.asarray(device_result["energy"]) - energy).max()),
                orthonormality=float(np.abs(Qd.T @ Qd - np.eye(Qd.shape[1])).max()),
                note="device POD (one of the timed steps) vs scipy dgesvd on the same 1e6 x 512 matrix; sigma_i below "
                     "sqrt(eps) sigma_1 carry eps sigma_1^2 / sigma_i from the Gram route (the test bar)")
    return out


def secondary_online_sweep(ctx, device, nt, n_mu, r, n_h, with_cpu):
    """BASELINE.json configs[4]: nt BDF2 steps x n_mu parameter points, r reduced DoFs.  One "step" = one time step of
    all n_mu reduced systems; value = nt * n_mu / wall with every table resident in HBM."""
    from romtime_amd import ops
    from romtime_amd.sweep import hrom_bdf_sweep, rom_bdf_sweep
    from romtime_amd.testing.workloads import c5_hyper_reduced

    terms, d, V, mus = c5_hyper_reduced(N=n_h, r=r, n_mu=n_mu, nt=nt)
    nnz = int(d["mass"].size)
    Vd = ops.to_device(V)
    dev = lambda a: ops.to_device(np.ascontiguousarray(a))
    direct_args = [Vd, d["indptr"], d["indices"], dev(d["mass"]), dev(d["terms"]), dev(d["term_coef"]), dev(d["tril"]),
                   dev(d["rhs_terms"]), dev(d["rhs_coef"]), d["dt"]]
    up = lambda term: dict(term, F=dev(term["F"]))
    h_args = (up(terms["mass"]), [up(t) for t in terms["lin"]], terms["nl"], [up(t) for t in terms["rhs"]], terms["dt"])

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    short = list(direct_args)
    short[5], short[8] = direct_args[5][:50], direct_args[8][:50]      # the (step, mu) coefficient tables
    rom_bdf_sweep(*short, bdf2=True)                                   # warm-up: arenas sized, kernels loaded
    wall_direct, uN_direct = timed(lambda: rom_bdf_sweep(*direct_args, bdf2=True))
    stats_direct = ctx.sweep_stats()
    hrom_bdf_sweep(*h_args, bdf2=True)
    wall_h, uN_h = timed(lambda: hrom_bdf_sweep(*h_args, bdf2=True))
    stats_h = ctx.sweep_stats()
    agree = float((uN_h - uN_direct).norm() / uN_direct.norm())       # the expansions represent the same model exactly
    # dominant kernel of the direct path, launched as the sweep launches it (n_mu value vectors on the pattern),
    # bracketed by the ctx's HIP event pair on its own stream
    ip, ix = ops.to_device_index(d["indptr"]), ops.to_device_index(d["indices"])
    kv = (direct_args[3][:, None] + 1e-4 * torch.randn((nnz, n_mu), dtype=torch.float64, device=device)).T.contiguous().T
    ctx.set_profile(True)
    ks = []
    for _ in range(10):
        for _ in range(10):                      # back to back, as in the sweep: the pair read is the last launch's
            ops.project_csr_batched(ip, ix, kv, Vd)
        ks.append(ctx.last_gemm_ms())
    ctx.set_profile(False)
    k_ms = float(np.mean(ks[2:]))
    flops = n_mu * (2.0 * nnz * r + 2.0 * n_h * r * r)
    alg_bytes = 8.0 * (n_mu * nnz + n_h * r)
    achieved = flops / (k_ms * 1e-3) / 1e12
    # HBM bytes per launch of the projection kernel: NOT measured in this run (PMC counters need rocprofv3's own passes);
    # read from the committed summary of such passes on this shape and labelled as static
    proj_traffic, proj_traffic_source = None, None
    tfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "project_traffic.json")
    if os.path.exists(tfile) and n_mu == 32 and r == 80 and n_h == 100_000:
        with open(tfile) as fp:
            tj = json.load(fp)
        proj_traffic = tj.get("hbm_bytes_per_launch")
        proj_traffic_source = f"static: profiles/project_traffic.json ({tj.get('source', '')})"
    out = {
        "metric": "reduced timesteps/s",
        "config": {"workload": f"online_sweep_{nt}steps_x_{n_mu}mu_r{r}_bdf2", "n_h_direct_path": n_h, "nnz": nnz,
                   "interpolation_coefficients": int(sum(t["F"].shape[2] for t in [terms["mass"]] + terms["lin"])
                                                     + terms["nl"]["W"].shape[0])},
        "direct": {"value": nt * n_mu / wall_direct, "unit": "reduced timesteps/s", "wall_s": wall_direct,
                   "ms_per_step_all_mu": 1e3 * wall_direct / nt, "solves": stats_direct},
        "hyper_reduced": {"value": nt * n_mu / wall_h, "unit": "reduced timesteps/s", "wall_s": wall_h,
                          "ms_per_step_all_mu": 1e3 * wall_h / nt, "solves": stats_h,
                          "rel_l2_vs_direct_path": agree, "bound": "latency (2 dependent launches per step: expansion GEMM ~16 us, solve kernel ~22 us - the carried inverse used as a preconditioner of a few refinement steps, refreshed by Newton-Schulz only when they stop contracting fast: see solves.newton_iterations - and ~5 us between them)"},
        "roofline": dict(bound="mfma", achieved=achieved, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                         frac=achieved / FP64_MFMA_PEAK_TFLOPS, traffic=proj_traffic, traffic_source=proj_traffic_source,
                         kernel="project_fused_kernel<5,false> (V^T(A_b V) for the n_mu operators of one step; the event pair also covers the <5,true> launch, which returns at once for a banded pattern)", kernel_ms=k_ms,
                         algorithmic_flops=flops, algorithmic_bytes=alg_bytes,
                         whole_step_frac=(nt * flops / wall_direct / 1e12) / FP64_MFMA_PEAK_TFLOPS),
    }
    if with_cpu:
        from oracle import romtime_oracle as oracle
        from romtime_amd.testing.mock import AffineBurgers
        from threadpoolctl import threadpool_limits

        cores = min(16, len(os.sched_getaffinity(0)))
        with threadpool_limits(limits=cores):
            n_direct = 8
            small = AffineBurgers(N=n_h, nt=n_direct, dt=d["dt"], bdf2=True, seed=5)
            t0 = time.perf_counter()
            ref, _ = oracle.rom_solve_nonlinear(small, V, mus[0], solver=oracle.reduced_solve)   # GMRES as the reference
            cpu_direct = (time.perf_counter() - t0) / n_direct
            err_direct = float(np.linalg.norm(uN_direct[0, :n_direct].cpu().numpy().T - ref) / np.linalg.norm(ref))
            # the same loop with an exact dense solver: the reference's GMRES stops at a 1e-10 residual (rom.py:36) and is
            # itself ~1e-7 away from the solution of its own systems here, so THIS is the distance the 1e-10 bar is about
            exact, _ = oracle.rom_solve_nonlinear(small, V, mus[0], solver=np.linalg.solve)
            err_exact = float(np.linalg.norm(uN_direct[0, :n_direct].cpu().numpy().T - exact) / np.linalg.norm(exact))
            n_h_steps = 2000
            cut = lambda term: dict(term, F=term["F"][:n_h_steps])
            t0 = time.perf_counter()
            href = oracle.hrom_solve(cut(terms["mass"]), [cut(t) for t in terms["lin"]], terms["nl"],
                                     [cut(t) for t in terms["rhs"]], 0, r, n_h_steps, d["dt"], True)
            cpu_h = (time.perf_counter() - t0) / n_h_steps
            err_h = float(np.linalg.norm(uN_h[0, :n_h_steps].cpu().numpy().T - href) / np.linalg.norm(href))
        out["cpu_baseline"] = {
            "direct": dict(value=1.0 / cpu_direct, unit="reduced timesteps/s", cores=cores, kind="port",
                           sample=f"oracle.rom_solve_nonlinear (5 csr.dot + matmul projections + GMRES per step, rom.py:877-929) "
                                  f"for 1 mu x {n_direct} steps of the workload", rel_l2_device_vs_oracle=err_direct,
                           rel_l2_device_vs_exact_solver_oracle=err_exact),
            "hyper_reduced": dict(value=1.0 / cpu_h, unit="reduced timesteps/s", cores=cores, kind="port",
                                  sample=f"oracle.hrom_solve (theta solves + dense solve per step, deim.py:416-452) for 1 mu x "
                                         f"{n_h_steps} steps of the workload", rel_l2_device_vs_oracle=err_h),
        }
    return out


def free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n_ranks):
    """Start `n_ranks` copies of this script (one per GPU: RANK = LOCAL_RANK = i, rendezvous on 127.0.0.1) and wait for
    them; returns the exit code.  Called BEFORE anything initialises the GPU in this process - a process that has must
    neither exec nor be forked from.  A rank that dies takes the others with it (by PID, never by pattern)."""
    import signal
    import subprocess

    have = torch.cuda.device_count()          # counting devices does not initialise the runtime
    backend = os.environ.get("ROMTIME_BENCH_BACKEND", "nccl")
    if backend == "nccl" and not os.environ.get("ROMTIME_BENCH_DRYRUN") and have < n_ranks:
        print(f"bench.py: --gpus {n_ranks} but {have} GPU(s) visible (RCCL needs one GPU per rank; "
              f"ROMTIME_BENCH_BACKEND=gloo rehearses with shared GPUs)", file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for i in range(n_ranks):
        env = dict(os.environ, RANK=str(i), LOCAL_RANK=str(i), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n_ranks))
    while alive:
        for i in sorted(alive):
            code = procs[i].poll()
            if code is None:
                continue
            alive.discard(i)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                print(f"bench.py: rank {i} exited with {code}; stopping the other ranks", file=sys.stderr)
                for j in alive:
                    procs[j].send_signal(signal.SIGTERM)
        if alive:
            time.sleep(0.05)
    return rc


def dry_run_line(args, world, rank):
    """ROMTIME_BENCH_DRYRUN=1: the launch path, the rendezvous, the shard arithmetic and the shape of the line WITHOUT
    any computation - no GPU is touched, `value` is null and the line says so.  What tests/test_bench_launch_cpu.py
    runs; never a measurement."""
    import torch.distributed as dist

    group_world = world
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    n_chunks = (args.rows + CHUNK_ROWS - 1) // CHUNK_ROWS
    per_rank = (n_chunks + world - 1) // world
    n_local = sum(min(CHUNK_ROWS, args.rows - c * CHUNK_ROWS)
                  for c in range(rank * per_rank, min(n_chunks, (rank + 1) * per_rank)))
    rows = torch.tensor([float(n_local)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(rows)
        group_world = dist.get_world_size()
    if rank == 0:
        assert int(rows.item()) == args.rows, "the row shards do not add up to the global matrix"
        line = base_line(args, group_world, n_local, value=None, ms_per_step=None, passes=None, mode="dry-run",
                         latency_ms=None)
        line["dry_run"] = True
        line["fallback_counters"] = dict.fromkeys(FALLBACK_COUNTERS)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


FALLBACK_COUNTERS = ("eig_timeouts", "eig_general_form", "eig_one_xcd", "sets_recomputed")


def base_line(args, world, n_local, value, ms_per_step, passes, mode, latency_ms):
    n_h, n, r = args.rows, args.cols, args.modes
    return {
        "metric": "snapshot-DoF/s for POD build",
        "value": value,
        "unit": "snapshot-DoF/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"pod_{n_h}x{n}_r{r}_normalize", "n_h": n_h, "n_snapshots": n, "modes": r,
                   "rows_per_gpu": n_local, "passes": passes, "parallelism": f"row-sharded x{world}",
                   "mode": mode, "single_pod_latency_ms": latency_ms},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=N_H)
    ap.add_argument("--cols", type=int, default=N_S)
    ap.add_argument("--modes", type=int, default=R_MODES)
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the online-sweep half of the metric (N = 1 only)")
    ap.add_argument("--sweep-steps", type=int, default=10_000)
    ap.add_argument("--no-latency", action="store_true",
                    help="pipeline mode: skip the 8 single-POD (latency mode) runs after the timed region - used under "
                         "rocprofv3 so that the kernel statistics are those of the timed region's kernels only")
    ap.add_argument("--mode", choices=("auto", "pipeline", "latency"), default="auto",
                    help="pipeline: the steps run through PodPipeline (eigensolve of step i beside the Gram of step i+1 on "
                         "CU-partitioned streams); latency: one pod_device call after the other; auto = pipeline")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This process has not touched the
        # GPU (importing torch does not) and never will: it only waits for its children.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a line whose n_gpus differs from what was asked for would be read as the N-GPU number
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to run")
    dry_run = bool(os.environ.get("ROMTIME_BENCH_DRYRUN"))
    if dry_run:
        return dry_run_line(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: romtime_amd has no CPU path")
    # ROMTIME_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks then share
    # devices; RCCL refuses that).  The driver's runs use the default: nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("ROMTIME_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    group = None
    force_dist = world == 1 and bool(os.environ.get("ROMTIME_FORCE_COLLECTIVES"))
    if force_dist:
        # one rank, but through the process group and its RCCL calls: the rehearsal a one-GPU box allows
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    distributed = world > 1 or force_dist
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        group = dist.group.WORLD

    from romtime_amd import ops, pod
    from romtime_amd._lib import Context

    n_h, n, r = args.rows, args.cols, args.modes
    X = build_local_matrix(rank, world, n_h, n, device)
    n_local = X.shape[0]
    ctx = Context.current()
    if world > torch.cuda.device_count():
        ctx.set_option("eig_one_xcd", 0)  # rehearsal with ranks sharing a GPU: no rank can have a whole XCD

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    mode = args.mode if args.mode != "auto" else "pipeline"
    latency_ms = None
    whole_chip_gram_ms = None
    pipe, why_not = None, None
    if mode == "pipeline":
        # The K steps are K independent PODs (as the per-parameter PODs of a tree walk are): PodPipeline keeps two of
        # them in flight, the n x n eigensolve of one beside the Gram kernel of the next on disjoint CUs.  Every step
        # delivers its complete result (basis on the device, all singular values on the host) before the region ends.
        from romtime_amd.pipeline import PodPipeline

        try:
            if world > torch.cuda.device_count():
                # gloo rehearsal, ranks sharing a GPU: every rank's eigensolver team needs CUs of its own
                pipe = PodPipeline(group=group, eig_first_cu=4 * local_rank, gram_range=(4 * world, 32 - 4 * world))
            else:
                # ROMTIME_PIPELINE_EIG_CUS: experiments with the CU partition (CUs of every XCD given to the eigensolver stream)
                pipe = PodPipeline(group=group, eig_cus_per_xcd=int(os.environ.get("ROMTIME_PIPELINE_EIG_CUS", "4")))
        except Exception as exc:  # noqa: BLE001  (a host that refuses CU-masked queues: one POD after the other instead)
            if args.mode == "pipeline":
                raise
            why_not = repr(exc)
        ok = torch.tensor([0.0 if pipe is None else 1.0], dtype=torch.float64, device=device)
        if distributed:
            import torch.distributed as dist

            dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # every rank takes the same route
        if float(ok.item()) == 0.0:
            pipe, mode = None, f"latency (pipeline unavailable: {why_not})"
    if pipe is not None:
        pipe.map([X] * max(args.warmup, 1), num=r, normalize=True)
        pipe.gram_kernel_ms.clear()
        barrier()
        t0 = time.perf_counter()
        outs = pipe.map([X] * args.steps, num=r, normalize=True)
        barrier()
        elapsed = time.perf_counter() - t0
        out = dict(passes=outs[-1]["passes"])
        kept = dict(Q=outs[-1]["Q"], s=outs[-1]["s"], energy=outs[-1]["energy"])   # for cpu_baseline.parity
        gram_ms = list(pipe.gram_kernel_ms)
        stage_ms = dict(pipe.last_stage_ms, sets_recomputed_on_regular_route=float(pipe.recomputed),
                        eigensolver_cus=float(pipe.eig_cus))
        if world == 1 and not args.no_latency:   # one POD on its own (latency mode), for the record
            for _ in range(3):
                pod.pod_device(X, num=r, normalize=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                pod.pod_device(X, num=r, normalize=True)
            torch.cuda.synchronize()
            latency_ms = 1e3 * (time.perf_counter() - t1) / 5
            # the Gram kernel on the whole chip (what a single POD runs: one launch, paced - see DESIGN section 4), its
            # own HIP event pair on the ctx stream
            # (sustained: the event pair of the last of eight launches back to back - a launch on its own, or the first after
            # an eigensolve during which 7/8 of the chip idled, runs ~8 % longer at the clock it finds)
            ctx.set_profile(True)
            whole = []
            for _ in range(3):
                for _ in range(8):
                    ops.gram(X)
                torch.cuda.synchronize()
                whole.append(ctx.last_gram_ms())
            ctx.set_profile(False)
            whole_chip_gram_ms = float(np.mean(whole))
        del outs
        pipe.close()
        counts = [pipe.ctxE.counter(c) + ctx.counter(c) for c in FALLBACK_COUNTERS[:3]] + [pipe.recomputed]
        del pipe
        from romtime_amd import pipeline as _pipeline

        _pipeline.shutdown()   # every tensor the pipeline produced is gone: the CU-masked streams can go too
    else:
        def step():
            return pod.pod_device(X, num=r, normalize=True, group=group)

        for _ in range(args.warmup):
            out = step()
        ctx.set_profile(True)
        gram_ms = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
            # the Gram kernel's own HIP event pair on the ctx stream (rt_last_gram_ms), read at the end of the step
            # when the kernel has long finished: nothing in the timed region waits for the device except the
            # eigenvalue fetch that orth's return values need
            gram_ms.append(out.get("gram_kernel_ms", float("nan")))
        barrier()
        elapsed = time.perf_counter() - t0
        ctx.set_profile(False)
        stage_ms = dict(pod.stage_timings())   # stream events of the last step, resolved after the timed region
        kept = dict(Q=out["Q"], s=out["s"], energy=out["energy"])
        counts = [ctx.counter(c) for c in FALLBACK_COUNTERS[:3]] + [0]

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    # what would hide a slow step: eigensolver hand-off time-outs, contexts that fell back to the general hand-off form,
    # sets recomputed on the regular route - summed over the ranks (warm-up included) and printed in every line
    fb = torch.tensor(counts, dtype=torch.float64, device=device)
    if distributed:
        import torch.distributed as dist

        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(fb, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    fallback = {name: int(v) for name, v in zip(FALLBACK_COUNTERS, fb.tolist())}

    if rank == 0:
        pipe_mode = mode == "pipeline"
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_h * n * args.steps / elapsed
        k_ms = float(np.nanmean(gram_ms))
        alg_flops = n_local * n * (n + 1)           # symmetric Gram: N n (n+1) flops
        alg_bytes = 8 * (n_local * n + n * n)       # read X once + write G
        achieved = alg_flops / (k_ms * 1e-3) / 1e12
        # HBM bytes per Gram: NOT measured in this run - PMC counters need rocprofv3's own passes over this same
        # command; the committed summary of those passes is read and labelled as such
        traffic, traffic_source = None, None
        tfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "gram_traffic.json")
        if world == 1 and (n_h, n) == (N_H, N_S) and os.path.exists(tfile):
            tj = json.load(open(tfile))
            traffic = tj.get("hbm_bytes_per_gram")
            traffic_source = f"static: profiles/gram_traffic.json ({tj.get('source', 'rocprofv3 --pmc passes of `python3 bench.py`')})"
        roofline = dict(bound="mfma", achieved=achieved, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=achieved / FP64_MFMA_PEAK_TFLOPS, traffic=traffic, traffic_source=traffic_source,
                        kernel="gram128_kernel<KC,false> + gram128_kernel<KC,true> (off-diagonal + diagonal tiles "
                               "of one Gram, one event pair around both launches"
                               + ("; in pipeline mode the pair is on the Gram's own CU-masked stream, includes the 0.03 ms "
                                  "slab reduction, and the kernels hold 224 of the 256 CUs)" if pipe_mode else ")"),
                        kernel_ms=k_ms, frac_of_cu_share_peak=(achieved / FP64_MFMA_PEAK_TFLOPS * 256.0 / 224.0
                                                                 if pipe_mode else achieved / FP64_MFMA_PEAK_TFLOPS),
                        algorithmic_flops=alg_flops, algorithmic_bytes=alg_bytes,
                        hbm_frac=alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        note="FP64 Gram at n=512 is 64 flop/B: matrix-core bound, not HBM bound.  peak = the spec figure "
                             "(2.4 GHz x 32 flop/clk/SIMD); this part sustains 47-48 TF on FP64 MFMAs alone with the pipe "
                             "saturated (profiles/r03_mfma_sustained.txt) and 66.7 TF in this kernel on all-zero data: "
                             "the kernel runs at the board's power limit")
        if whole_chip_gram_ms is not None and whole_chip_gram_ms == whole_chip_gram_ms:
            wc_traffic = tj.get("whole_chip", {}).get("hbm_bytes_per_gram") if traffic is not None else None
            wc = alg_flops / (whole_chip_gram_ms * 1e-3) / 1e12
            roofline["whole_chip_launch"] = dict(
                kernel="gram128_merged_kernel<KC> (the Gram of a POD on its own: all 256 CUs, one launch, workgroups of an "
                       "XCD paced so that every panel is fetched once)",
                kernel_ms=whole_chip_gram_ms, achieved=wc, frac=wc / FP64_MFMA_PEAK_TFLOPS, traffic=wc_traffic,
                traffic_source=(f"static: profiles/gram_traffic.json ({tj['whole_chip'].get('source')})"
                                if wc_traffic is not None else None))
        line = base_line(args, world, n_local, value, ms_per_step, out["passes"], mode, latency_ms)
        line["stage_ms"] = {k: round(v, 4) for k, v in stage_ms.items()}
        line["roofline"] = roofline
        line["fallback_counters"] = fallback
        if world == 1 and not args.no_cpu_baseline:
            rows = min(args.cpu_sample_rows, n_h)
            X_host = X[:rows].cpu().numpy()          # the workload's own matrix (device and host RNG streams differ)
            line["cpu_baseline"] = cpu_baseline(X_host, r, device_result=kept if rows == n_h else None)
            del X_host
        if world == 1 and not args.no_secondary:
            del X, out
            torch.cuda.empty_cache()
            line["secondary"] = secondary_online_sweep(ctx, device, nt=args.sweep_steps, n_mu=32, r=80, n_h=100_000,
                                                       with_cpu=not args.no_cpu_baseline)
        print(json.dumps(line), flush=True)
    if distributed:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
