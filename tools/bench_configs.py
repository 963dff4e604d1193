#!/usr/bin/env python3
"""Secondary measurements on the BASELINE.json configs that bench.py does not headline
(C2 POD 1e5x256, C4 MDEIM greedy + project_basis, C5 online reduced steps), each next to the oracle
on a bounded sample.  Prints one JSON object per config; results go to BASELINE.md / profiles/."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import romtime_oracle as oracle  # noqa: E402
from romtime_amd import ops, pod  # noqa: E402
from scipy.sparse import csr_matrix  # noqa: E402

HBM, MFMA = 8000.0, 78.6


def ev_time(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def wall_time(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, out


def penta(N, rng):
    offs = [-2, -1, 0, 1, 2]
    rows = np.concatenate([np.arange(max(0, -o), min(N, N - o)) for o in offs])
    cols = np.concatenate([np.arange(max(0, -o), min(N, N - o)) + o for o in offs])
    A = csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(N, N))
    A.sort_indices()
    return A


def c2():
    N, n, r = 100_000, 256, 40
    g = torch.Generator(device="cuda").manual_seed(2)
    s = torch.from_numpy(10.0 ** (-6.0 * np.arange(n) / (n - 1))).cuda()
    V0, _ = torch.linalg.qr(torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g))
    X = (torch.randn((N, n), dtype=torch.float64, device="cuda", generator=g) / np.sqrt(N)) @ (s[:, None] * V0.T)
    ms, out = wall_time(lambda: pod.pod_device(X, num=r, normalize=True), reps=5)
    Xh = X.cpu().numpy()
    t0 = time.perf_counter()
    Qr, sr, er = oracle.orth(Xh, num=r, normalize=True)
    cpu = time.perf_counter() - t0
    Q = out["Q"].cpu().numpy()
    sub = np.linalg.norm(Q @ (Q.T @ Qr) - Qr, 2)
    return dict(config="C2 POD 1e5x256 r40", gpu_ms=ms, dof_per_s=N * n / ms * 1e3, cpu_s=cpu, cpu_dof_per_s=N * n / cpu,
                sigma_rel_err=float(np.abs(out["s"][:r] - sr[:r]).max() / sr[0]), subspace_dist=float(sub),
                passes=out["passes"])


def c2pipe(sets=16):
    """The tree walks' pattern (rom.py:317-406, deim.py:279-397): many independent small PODs.  The same C2-sized sets
    one after the other and through the CU-partitioned pipeline (eigensolve of set i beside the Gram of set i+1)."""
    from romtime_amd import pipeline

    N, n, r = 100_000, 256, 40
    g = torch.Generator(device="cuda").manual_seed(2)
    s = torch.from_numpy(10.0 ** (-6.0 * np.arange(n) / (n - 1))).cuda()
    Xs = []
    for _ in range(sets):
        V0, _ = torch.linalg.qr(torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g))
        Xs.append((torch.randn((N, n), dtype=torch.float64, device="cuda", generator=g) / np.sqrt(N)) @ (s[:, None] * V0.T))
    ms_seq, outs = wall_time(lambda: [pod.pod_device(X, num=r, normalize=True) for X in Xs], reps=3)
    pipe = pipeline.PodPipeline(small_set=0)     # forced onto the streams; by default such small sets take the regular route
    ms_pipe, pouts = wall_time(lambda: pipe.map(Xs, num=r, normalize=True), reps=3)
    lanes = pipeline.PodLanes()
    ms_lanes, louts = wall_time(lambda: lanes.map(Xs, num=r, normalize=True), reps=3)
    worst_l = 0.0
    for a, b in zip(outs, louts):
        Qa, Qb = a["Q"], b["Q"]
        worst_l = max(worst_l, float(torch.linalg.matrix_norm(Qb @ (Qb.T @ Qa) - Qa, 2)))
    worst = 0.0
    for a, b in zip(outs, pouts):
        Qa, Qb = a["Q"], b["Q"]
        worst = max(worst, float(torch.linalg.matrix_norm(Qb @ (Qb.T @ Qa) - Qa, 2)))
    res = dict(config=f"C2 x {sets}: independent PODs of 1e5x256 r40, one after the other / PodPipeline.map (forced) / PodLanes.map",
               sequential_ms_per_pod=ms_seq / sets, forced_pipeline_ms_per_pod=ms_pipe / sets,
               sequential_dof_per_s=N * n * sets / ms_seq * 1e3, lanes_ms_per_pod=ms_lanes / sets,
               lanes_dof_per_s=N * n * sets / ms_lanes * 1e3, lanes_recomputed=int(lanes.recomputed),
               subspace_dist_lanes_vs_sequential=worst_l, recomputed=int(pipe.recomputed),
               subspace_dist_pipeline_vs_sequential=worst)
    pipeline.shutdown()
    return res


def c4():
    N, nnz_row, n_ops, m, r = 100_000, 5, 200, 120, 80
    rng = np.random.RandomState(4)
    A = penta(N, rng)
    nnz = A.nnz
    # 200 value vectors = smooth combinations of 8 spatial profiles + 1e-6 noise, row 0 zeroed
    x = np.linspace(0, 1, nnz)
    B = np.stack([np.sin((q + 1) * np.pi * x) * (1 + 0.1 * q) for q in range(8)], axis=1)
    theta = rng.standard_normal((8, n_ops))
    S = torch.from_numpy(B @ theta + 1e-6 * rng.standard_normal((nnz, n_ops))).cuda()
    S[0, :] = 0.0
    ms_pod, out = wall_time(lambda: pod.pod_device(S, num=m, normalize=False), reps=2)
    Phi = out["Q"]
    ms_greedy, g = wall_time(lambda: ops.deim_greedy(Phi, want_margin=False), reps=2)
    greedy_bytes = 8.0 * nnz * sum(k + 2 for k in range(m))      # reference algorithm: step k reads k+1 columns
    L = 8                                                        # blocked elimination actually moves about this much:
    # m^2/(2L) columns read by the block sweeps, 2m for their phi columns in / t columns out, (L+3)/2 per column launch
    greedy_bytes_blocked = 8.0 * nnz * (m * m / (2.0 * L) + 2 * m + m * (L + 3) / 2.0)
    V, _ = torch.linalg.qr(torch.randn((N, r), dtype=torch.float64, device="cuda"))
    ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
    ms_proj, AN = wall_time(lambda: ops.project_csr_batched(ip, ix, Phi, V), reps=2)
    proj_flops = m * (2.0 * nnz * r + 2.0 * N * r * r)
    # oracle on a bounded sample: greedy on the first 24 modes, projection of 4 modes
    Ph = Phi.cpu().numpy()
    t0 = time.perf_counter()
    dofs, _, _ = oracle.deim_greedy(Ph[:, :24])
    cpu_greedy24 = time.perf_counter() - t0
    Vh = V.cpu().numpy()
    t0 = time.perf_counter()
    ref4 = [oracle.project_csr(csr_matrix((Ph[:, i], A.indices, A.indptr), shape=(N, N)), Vh) for i in range(4)]
    cpu_proj4 = time.perf_counter() - t0
    idx = g[0].cpu().numpy()
    return dict(config="C4 MDEIM 200 ops N=1e5 nnz~5e5 m=120 r=80", pod_ms=ms_pod, greedy_ms=ms_greedy,
                greedy_GBps_reference_algorithm_bytes=greedy_bytes / ms_greedy / 1e6,
                greedy_GBps_moved=greedy_bytes_blocked / ms_greedy / 1e6,
                greedy_hbm_frac=greedy_bytes_blocked / ms_greedy / 1e6 / HBM,
                project_ms=ms_proj, project_TFs=proj_flops / ms_proj / 1e9, project_mfma_frac=proj_flops / ms_proj / 1e9 / MFMA,
                cpu_greedy_first24_s=cpu_greedy24, greedy_first24_match=bool(list(idx[:24]) == list(dofs)),
                cpu_project_4modes_s=cpu_proj4,
                project_err=float(max(np.abs(AN[i].cpu().numpy() - ref4[i]).max() for i in range(4))))


def c5(nt=200, n_mu=32):
    """Online direct path per (mu, step): 2 projections (M and the pre-summed K) + r x r solve + lift."""
    N, r = 100_000, 80
    rng = np.random.RandomState(5)
    A = penta(N, rng)
    ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
    V, _ = torch.linalg.qr(torch.randn((N, r), dtype=torch.float64, device="cuda"))
    Mdata = torch.from_numpy(np.abs(A.data) + 3.0).cuda()
    Kdata = torch.randn((A.nnz, n_mu), dtype=torch.float64, device="cuda") * 0.1
    Kdata += Mdata[:, None]
    f = torch.randn(N, dtype=torch.float64, device="cuda")

    def step():
        MN = ops.project_csr(ip, ix, Mdata, V)
        KN = ops.project_csr_batched(ip, ix, Kdata, V)              # (n_mu, r, r)
        bN = ops.gemm_tn(V, f)
        x, info = ops.dense_solve(KN, bN.unsqueeze(0).expand(n_mu, r))
        return ops.gemm_nn(V, x.T.contiguous())                      # lift all n_mu solutions

    ms, _ = wall_time(step, reps=nt // 10, warm=2)
    flops = n_mu * 2 * (2.0 * A.nnz * r + 2.0 * N * r * r)
    # oracle: one (mu, step)
    Vh, Ah = V.cpu().numpy(), A
    t0 = time.perf_counter()
    for _ in range(3):
        MNh = oracle.project_csr(Ah, Vh)
        KNh = oracle.project_csr(Ah, Vh) + np.eye(r)
        u = oracle.reduced_solve(KNh, Vh.T @ f.cpu().numpy())
        Vh @ u
    cpu = (time.perf_counter() - t0) / 3
    return dict(config=f"C5 online r=80 N=1e5 {n_mu} mu per batched step (fused SpMM+MFMA projection)", ms_per_batched_step=ms,
                reduced_steps_per_s=n_mu / ms * 1e3, TFs=flops / ms / 1e9, mfma_frac=flops / ms / 1e9 / MFMA,
                cpu_s_per_step=cpu, cpu_steps_per_s=1.0 / cpu)


def c5sweep(nt=200, n_mu=32, N=100_000, r=80):
    """Config 5 as a device-resident sweep (rt_rom_bdf_sweep): nt BDF2 steps x n_mu parameter points."""
    from romtime_amd.sweep import rom_bdf_sweep
    from romtime_amd.testing.mock import AffineBurgers

    fom = AffineBurgers(N=N, nt=nt, dt=1e-4, bdf2=True, seed=5)
    xs = (np.arange(N) + 0.5) / N
    rng = np.random.RandomState(1)
    V, _ = np.linalg.qr(np.stack([np.sin((k + 1) * np.pi * xs) for k in range(r)], axis=1) + 1e-3 * rng.standard_normal((N, r)))
    mus = [dict(alpha=0.5 + 0.02 * i, beta=1.0 - 0.01 * i, delta=0.3 + 0.005 * i, omega=7.0 + 0.1 * i) for i in range(n_mu)]
    d = fom.descriptor(mus)
    args = [ops.to_device(V), d["indptr"], d["indices"], ops.to_device(d["mass"]), ops.to_device(d["terms"]),
            ops.to_device(d["term_coef"]), ops.to_device(d["tril"]), ops.to_device(d["rhs_terms"]),
            ops.to_device(d["rhs_coef"]), d["dt"]]
    rom_bdf_sweep(*args, bdf2=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    uN = rom_bdf_sweep(*args, bdf2=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    nnz = d["mass"].size
    flops = nt * n_mu * (2.0 * nnz * r + 2.0 * N * r * r)
    # oracle: the reference loop (5 separate projections per step, exact dense solve) for one mu, 3 steps
    small = AffineBurgers(N=N, nt=3, dt=1e-4, bdf2=True, seed=5)
    t0 = time.perf_counter()
    ref, _ = oracle.rom_solve_nonlinear(small, V, mus[0], solver=np.linalg.solve)
    cpu = (time.perf_counter() - t0) / 3
    err = np.linalg.norm(uN[0, :3].cpu().numpy().T - ref) / np.linalg.norm(ref)
    return dict(config=f"C5 online sweep r={r} N={N} {n_mu} mu x {nt} BDF2 steps on device (full config: 1e4 steps)",
                wall_s=wall, ms_per_step_all_mu=1e3 * wall / nt, reduced_steps_per_s=nt * n_mu / wall,
                TFs=flops / wall / 1e12, mfma_frac=flops / wall / 1e12 / MFMA, cpu_s_per_reduced_step=cpu,
                cpu_reduced_steps_per_s=1.0 / cpu, first3_rel_err_vs_oracle=float(err))


def c5h(nt=2000, n_mu=32, N=100_000, r=80, m_lin=40, m_nl=120, m_rhs=20):
    """Config 5 through the hyper-reduced path (SURVEY 8d "online step, hyper-reduced path"): every reduced operator
    an (M)DEIM expansion.  The expansions are built so that they represent the AffineBurgers model exactly (true
    modes + padding modes, coefficient tables rotated by a random PT_U), hence the device-resident direct sweep
    (c5sweep) is the full-size cross-check."""
    from romtime_amd.sweep import hrom_bdf_sweep, rom_bdf_sweep
    from romtime_amd.testing.mock import AffineBurgers

    nt_direct = 40
    fom = AffineBurgers(N=N, nt=nt, dt=1e-4, bdf2=True, seed=5)
    xs = (np.arange(N) + 0.5) / N
    rng = np.random.RandomState(1)
    V, _ = np.linalg.qr(np.stack([np.sin((k + 1) * np.pi * xs) for k in range(r)], axis=1) + 1e-3 * rng.standard_normal((N, r)))
    mus = [dict(alpha=0.5 + 0.02 * i, beta=1.0 - 0.01 * i, delta=0.3 + 0.005 * i, omega=7.0 + 0.1 * i) for i in range(n_mu)]
    d = fom.descriptor(mus)
    Vd = ops.to_device(V)
    ip, ix = ops.to_device_index(d["indptr"]), ops.to_device_index(d["indices"])
    proj = lambda vals: ops.project_csr(ip, ix, ops.to_device(vals), Vd).cpu().numpy().reshape(-1)
    rr_of = np.repeat(np.arange(N), np.diff(d["indptr"]))

    def embed(true_cols, coefs, m):
        """true_cols (r^2 x k) with coefficient table coefs (nt x n_mu x k) -> an m-mode term with a random PT_U."""
        k = true_cols.shape[1]
        basis_rom = np.concatenate([true_cols, 1e-3 * rng.standard_normal((true_cols.shape[0], m - k))], axis=1)
        PT_U, _ = np.linalg.qr(rng.standard_normal((m, m)))
        theta = np.concatenate([coefs, np.zeros(coefs.shape[:-1] + (m - k,))], axis=-1)
        return dict(PT_U=PT_U, basis_rom=basis_rom, F=theta @ PT_U.T)

    ones = np.ones((nt, n_mu, 1))
    mass = embed(proj(d["mass"])[:, None], ones, m_lin)
    lin = [embed(proj(d["terms"][q])[:, None], d["term_coef"][:, :, q:q + 1], m_lin) for q in range(3)]
    # trilinear: V^T diag(V u) T V = sum_k u_k N_k
    Nk = np.stack([proj(V[rr_of, k] * d["tril"]) for k in range(r)], axis=1)               # r^2 x r
    nl_full = embed(Nk, np.zeros((1, 1, r)), m_nl)
    nl = dict(PT_U=nl_full["PT_U"], basis_rom=nl_full["basis_rom"],
              W=nl_full["PT_U"] @ np.concatenate([np.eye(r), np.zeros((m_nl - r, r))], axis=0))
    fN = (V.T @ d["rhs_terms"].T)                                                            # r x F
    rhs = [embed(fN, d["rhs_coef"], m_rhs)]
    host_terms = (mass, lin, rhs)
    up = lambda term: dict(term, F=ops.to_device(np.ascontiguousarray(term["F"])))   # tables resident, as in bench.py
    mass, lin, rhs = up(mass), [up(t) for t in lin], [up(t) for t in rhs]
    args = (mass, lin, nl, rhs, d["dt"])
    hrom_bdf_sweep(*args, bdf2=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    uN = hrom_bdf_sweep(*args, bdf2=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ref = rom_bdf_sweep(Vd, d["indptr"], d["indices"], ops.to_device(d["mass"]), ops.to_device(d["terms"]),
                        ops.to_device(d["term_coef"][:nt_direct]), ops.to_device(d["tril"]), ops.to_device(d["rhs_terms"]),
                        ops.to_device(d["rhs_coef"][:nt_direct]), d["dt"], bdf2=True)
    cross = float((uN[:, :nt_direct] - ref).norm() / ref.norm())
    t0 = time.perf_counter()
    small = 5
    cut = lambda term: dict(term, F=term["F"][:small])
    hm, hl, hr = host_terms
    oref = oracle.hrom_solve(cut(hm), [cut(t) for t in hl], nl, [cut(t) for t in hr], 0, r, small, d["dt"], True)
    cpu = (time.perf_counter() - t0) / small
    err = float(np.linalg.norm(uN[0, :small].cpu().numpy().T - oref) / np.linalg.norm(oref))
    M = m_lin * 4 + m_nl
    return dict(config=f"C5 hyper-reduced sweep r={r} {n_mu} mu x {nt} BDF2 steps, {M} interpolation coefficients "
                       f"(mass/stiffness/convection/nonlinear-lifting {m_lin} each, trilinear {m_nl}, rhs {m_rhs})",
                wall_s=wall, ms_per_step_all_mu=1e3 * wall / nt, reduced_steps_per_s=nt * n_mu / wall,
                cpu_s_per_reduced_step=cpu, cpu_reduced_steps_per_s=1.0 / cpu, first5_rel_err_vs_oracle=err,
                rel_diff_vs_direct_device_sweep_first40=cross)


def walk16(n_mu=16, N=100_000, n_t=256, num_t=40, num_mu=40):
    """A DEIM tree walk through the class surface (deim.py:279-397): 16 parameters x (1e5 x 256) time-level snapshot
    sets, 40 modes each, then the mu-level POD of the 1e5 x 640 concatenation - the device-resident walk (each set
    uploaded once, time-level PODs through the POD lanes, bases concatenated on the device; walks.py) against the
    reference's loop as round 2 ran it (one orth(ndarray) per parameter: upload, POD, download; np.hstack; orth again).
    The FOM callback is a table lookup, so what is timed is the walk, not an assembly."""
    from romtime_amd import DiscreteEmpiricalInterpolation
    from romtime_amd.conventions import Stage

    rng = np.random.RandomState(16)
    sig = 10.0 ** (-6.0 * np.arange(n_t) / (n_t - 1))
    V0, _ = np.linalg.qr(rng.standard_normal((n_t, n_t)))
    tables = {}
    for j in range(n_mu):                      # snapshot j of parameter i = row j of tables[i] (contiguous vectors)
        Z = torch.randn((N, n_t), dtype=torch.float64, device="cuda") / np.sqrt(N)
        tables[j] = (Z @ torch.from_numpy((sig * (1.0 + 0.05 * j))[:, None] * V0.T).cuda()).T.contiguous().cpu().numpy()
    ts = list(range(n_t))
    space = [dict(i=j) for j in range(n_mu)]
    assemble = lambda mu, t: tables[mu["i"]][t]

    def run(host_loop):
        d = DiscreteEmpiricalInterpolation(assemble=assemble, name="walk16")
        d.setup()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if host_loop:
            basis, sig_mu = d._tree_walk_host(space, ts, False, num_mu, num_t, None, None)
        else:
            basis, sig_mu = d.tree_walk(ts=ts, normalize=False, num_mu=num_mu, num_t=num_t, mu_space=space)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, basis, sig_mu, d.report[Stage.OFFLINE]

    run(False), run(True)                      # warm-up: lanes, arenas, pinned buffers
    t_dev, B_dev, s_dev, rep_dev = run(False)
    t_host, B_host, s_host, rep_host = run(True)
    sub = np.linalg.norm(B_dev - B_host @ (B_host.T @ B_dev), 2)
    # device-resident sets (no FOM, no PCIe): what the lanes and the on-device concatenation are worth by themselves
    from romtime_amd import walks

    dev_sets = [ops.to_device(tables[j].T) for j in range(n_mu)]

    def lanes():
        per = [o["Q"] for o in walks.pod_sequence(dev_sets, num=num_t, normalize=False)]
        return walks.pod_of_stack(per, num=num_mu, normalize=False)

    def one_by_one():
        per = [pod.pod_device(X, num=num_t, normalize=False)["Q"] for X in dev_sets]
        return pod.pod_device(torch.cat(per, dim=1), num=num_mu, normalize=False)

    ms_lanes, _ = wall_time(lanes, reps=3)
    ms_seq, _ = wall_time(one_by_one, reps=3)
    walks.MODE = "workers"
    ms_workers, _ = wall_time(lanes, reps=3)
    walks.MODE = "auto"
    return dict(config=f"tree walk, {n_mu} mu x ({N} x {n_t}), {num_t} modes per mu, {num_mu} final (DEIM class surface)",
                device_walk_s=t_dev, host_loop_s=t_host, speedup=t_host / t_dev,
                sigma_mu_rel_diff=float(np.abs(s_dev - s_host).max() / s_host[0]), subspace_dist=float(sub),
                basis_after_walk=rep_dev["basis-shape-after-tree-walk"] if "basis-shape-after-tree-walk" in rep_dev else None,
                device_resident_sets=dict(lanes_ms=ms_lanes, workers_ms=ms_workers, one_by_one_ms=ms_seq, speedup=ms_seq / ms_lanes),
                note="host-resident snapshots: per 205 MB set the round-2 loop spends 21 ms in np.array(list).T (page faults of "
                     "the fresh allocation), 4 ms on the upload, downloads every time-level basis and re-uploads the 512 MB "
                     "concatenation; the device walk stacks the FOM's vectors straight into reusable pinned memory (9 ms), "
                     "sends them asynchronously and keeps the bases on the device")


def walkdeep(n_mu=8, N=100_000, n_t=256, tol_t=1.0 - 1e-8, num_mu=40):
    """The tree walk with the truncation the reference's drivers use: an energy tolerance at the time level (rom.py:335;
    tol = 1 - 1e-8 keeps modes down to ~1e-4 sigma_1, so every time-level POD needs deflated levels).  Device-resident
    sets; the three ways a sequence of such PODs can run: one after the other, through the enqueue-ahead lanes (every
    set turns out deep and is recomputed one after the other), through the worker threads (the regular route of every set
    in a thread, stream and context of its own)."""
    from romtime_amd import walks

    rng = np.random.RandomState(17)
    sig = 10.0 ** (-16.0 * np.arange(n_t) / (n_t - 1))
    V0, _ = np.linalg.qr(rng.standard_normal((n_t, n_t)))
    mix = torch.from_numpy(sig[:, None] * V0.T).cuda()
    sets = []
    for j in range(n_mu):
        U, _ = torch.linalg.qr(torch.randn((N, n_t), dtype=torch.float64, device="cuda"))
        sets.append((U @ mix) * (1.0 + 0.05 * j))
        del U

    def run(mode):
        def go():
            if mode == "sequential":
                per = [pod.pod_device(X, tol=tol_t, normalize=True) for X in sets]
            else:
                walks.MODE = mode
                per = list(walks.pod_sequence(sets, tol=tol_t, normalize=True))
                walks.MODE = "auto"
            top = walks.pod_of_stack([o["Q"] for o in per], num=num_mu, normalize=False)
            return per, top
        return wall_time(go, reps=3)

    out = {}
    ref = None
    for mode in ("sequential", "lanes", "workers"):
        ms, (per, top) = run(mode)
        out[mode + "_ms"] = ms
        if ref is None:
            ref = (per, top)
        else:
            assert [o["r"] for o in per] == [o["r"] for o in ref[0]], mode
            out[mode + "_sigma_rel_diff"] = float(max(np.abs(a["s"] - b["s"]).max() / b["s"][0] for a, b in zip(per, ref[0])))
    return dict(config=f"tree walk, deep time-level spectra: {n_mu} mu x ({N} x {n_t}), tol_t = 1 - 1e-8 "
                       f"({[o['r'] for o in ref[0]][:3]}... modes, {ref[0][0]['passes']} route), {num_mu} final; device-resident sets",
                **out, speedup_workers=out["sequential_ms"] / out["workers_ms"])


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c2pipe", "walk16", "walkdeep", "c4", "c5", "c5sweep", "c5h"]
    for w in which:
        print(json.dumps({"c2": c2, "c2pipe": c2pipe, "walk16": walk16, "walkdeep": walkdeep, "c4": c4, "c5": c5, "c5sweep": c5sweep, "c5h": c5h}[w]()), flush=True)
