"""Device-resident online sweep: the BDF time loop of ``RomConstructor*.solve`` (rom/rom.py:430-555,
direct path of :877-929) for many parameter points at once, without returning to the host between
steps (``rt_rom_bdf_sweep``).

The operators must be affine in fixed value vectors on one CSR pattern,
``K(mu, t, u*) = bdf M + dt (sum_q theta_q(mu, t) A_q + diag(u*) T)``, which is what a closed-form
assembly (or an MDEIM expansion in FOM coordinates) provides; the coefficient tables theta / phi are
evaluated on the host for all (step, mu) beforehand and uploaded once."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import ops
from ._lib import Context, SweepDesc

_p = C.c_void_p


def rom_bdf_sweep(V, indptr, indices, mass, terms, term_coef, tril, rhs_terms, rhs_coef, dt, bdf2=True):
    """Returns the reduced trajectories ``uN`` as a (n_mu, nt, r) CUDA tensor.

    V (N x r); indptr/indices CSR pattern; mass (nnz); terms (Q x nnz); term_coef (nt x n_mu x Q);
    tril (nnz) or None; rhs_terms (F x N); rhs_coef (nt x n_mu x F)."""
    ctx = Context.current()
    Vd = ops.to_device(np.ascontiguousarray(V) if isinstance(V, np.ndarray) else V).contiguous()
    N, r = Vd.shape
    ip, ix = ops.to_device_index(indptr), ops.to_device_index(indices)
    dev = lambda a: None if a is None else ops.to_device(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a).contiguous()
    mass_d, terms_d, tcoef_d, tril_d = dev(mass), dev(terms), dev(term_coef), dev(tril)
    rhs_d, rcoef_d = dev(rhs_terms), dev(rhs_coef)
    nt, n_mu = (tcoef_d.shape[0], tcoef_d.shape[1]) if tcoef_d is not None else (rcoef_d.shape[0], rcoef_d.shape[1])
    out = torch.empty((n_mu, nt, r), dtype=torch.float64, device=Vd.device)
    ptr = lambda t: _p(t.data_ptr()) if t is not None else _p(None)
    desc = SweepDesc(N=N, nnz=mass_d.numel(), r=r, n_mu=n_mu, nt=nt, dt=float(dt), bdf2=int(bool(bdf2)),
                     indptr=ptr(ip), indices=ptr(ix), V=ptr(Vd), mass_values=ptr(mass_d),
                     n_terms=0 if terms_d is None else terms_d.shape[0], term_values=ptr(terms_d), term_coef=ptr(tcoef_d),
                     tril_values=ptr(tril_d), n_rhs=0 if rhs_d is None else rhs_d.shape[0], rhs_terms=ptr(rhs_d),
                     rhs_coef=ptr(rcoef_d))
    ctx.check(ctx.lib.rt_rom_bdf_sweep(ctx.handle, C.byref(desc), _p(out.data_ptr())), "rt_rom_bdf_sweep")
    return out


def hrom_bdf_sweep(mass, lin, nl, rhs, dt, bdf2=True):
    """Hyper-reduced online sweep on the device (``rt_hrom_bdf_sweep``): the time loop of
    ``RomConstructor*.solve`` with every reduced operator obtained by (M)DEIM interpolation,
    ``interpolate(which=ROM)`` (deim.py:416-452, mdeim.py:230-261), for all parameter points at once.

    ``mass``: one term; ``lin``: list of terms; ``rhs``: list of vector terms.  A term is a dict with ``PT_U``
    (m x m), ``basis_rom`` (r^2 x m, or r x m for vectors) - the attributes of a projected reductor - and ``F``
    (nt x n_mu x m): the operator's own entries at the reductor's interpolation entries for every step and
    parameter point (what ``assemble(mu, t, entries=dofs)`` returns).  ``nl``: None or a dict with ``PT_U``,
    ``basis_rom``, ``W`` (m x r), optional ``C`` (nt x n_mu x m) and ``S`` (nt x n_mu): entries = S (W u_N* + C).
    The theta solve is folded into the expansion once (Z = basis_rom PT_U^-1); the loop then never touches anything
    of size N_h.  Returns ``uN`` (n_mu, nt, r) as a CUDA tensor."""
    from ._lib import HSweepDesc

    ctx = Context.current()

    def fold(term):  # rows = the r x r (or r) arrays multiplied by each local entry: PT_U^T Z = basis_rom^T
        PT_U, basis_rom = np.asarray(term["PT_U"], dtype=np.float64), np.asarray(term["basis_rom"], dtype=np.float64)
        if PT_U.shape[0] > 128:   # beyond the LDS-resident factorisation (include/romtime_hip.h): host LAPACK
            return np.linalg.solve(PT_U.T, basis_rom.T)
        Zt, info = ops.dense_solve_multi(ops.to_device(np.ascontiguousarray(PT_U.T)), ops.to_device(np.ascontiguousarray(basis_rom.T)))
        if int(info.item()) != 0:
            raise np.linalg.LinAlgError("Singular matrix")           # what np.linalg.solve raises (deim.py:491-492)
        return Zt.cpu().numpy()

    blocks = [fold(mass)] + [fold(t) for t in lin] + ([fold(nl)] if nl is not None else [])
    rr = blocks[0].shape[1]
    r = int(round(np.sqrt(rr)))
    assert r * r == rr and all(bk.shape[1] == rr for bk in blocks), "matrix terms must be r^2 x m"
    def dev(a):  # tables may already live on the device (torch tensors)
        if isinstance(a, torch.Tensor):
            return a.to(dtype=torch.float64).contiguous() if a.is_cuda else ops.to_device(a.numpy())
        return ops.to_device(np.ascontiguousarray(a, dtype=np.float64))

    def cat(terms):
        tabs = [dev(t["F"]) for t in terms]
        return tabs[0] if len(tabs) == 1 else torch.cat(tabs, dim=2).contiguous()

    nt, n_mu, m_mass = tuple(mass["F"].shape)
    m_lin = sum(int(t["F"].shape[2]) for t in lin)
    m_nl = 0 if nl is None else int(nl["W"].shape[0])
    Z = dev(np.vstack(blocks))
    Fm = dev(mass["F"])
    Fl = cat(lin) if lin else None
    Zf = dev(np.vstack([fold(t) for t in rhs])) if rhs else None
    Ff = cat(rhs) if rhs else None
    m_rhs = 0 if Zf is None else Zf.shape[0]
    W = dev(nl["W"]) if nl is not None else None
    Cn = dev(nl["C"]) if nl is not None and nl.get("C") is not None else None
    Sn = dev(nl["S"]) if nl is not None and nl.get("S") is not None else None
    out = torch.empty((n_mu, nt, r), dtype=torch.float64, device=Z.device)
    ptr = lambda t: _p(t.data_ptr()) if t is not None else _p(None)
    desc = HSweepDesc(r=r, n_mu=n_mu, nt=nt, dt=float(dt), bdf2=int(bool(bdf2)), m_mass=m_mass, m_lin=m_lin, m_nl=m_nl,
                      m_rhs=m_rhs, Z=ptr(Z), Zf=ptr(Zf), F_mass=ptr(Fm), F_lin=ptr(Fl), F_rhs=ptr(Ff), W=ptr(W),
                      C_nl=ptr(Cn), S_nl=ptr(Sn))
    ctx.check(ctx.lib.rt_hrom_bdf_sweep(ctx.handle, C.byref(desc), _p(out.data_ptr())), "rt_hrom_bdf_sweep")
    return out


def hrom_terms_from_rom(rom, mus):
    """Term dictionaries for :func:`hrom_bdf_sweep` from a ``RomConstructorNonlinear`` whose operators are all
    hyper-reduced and projected (``add_hyper_reductor`` + ``project_reductors``): the tables hold what each
    reductor's ``assemble(mu, t, entries=dofs)`` returns at every step for every parameter point - the same calls
    the host loop ``rom.solve`` makes one by one (rom.py:877-929 through deim.py:429-433).

    The state-dependent operator must be an affine map of the state that does not depend on (mu, t) (a trilinear
    form on a reference mesh plus constant boundary entries), which is checked; other FOMs fill the ``C`` / ``S``
    fields of the ``nl`` term themselves."""
    fom = rom.fom
    nt, dt = fom.domain["nt"], fom.dt
    ts = dt * np.arange(1, nt + 1)

    def table(red):
        return np.array([[red._local_values(mu, t) for mu in mus] for t in ts])

    def term(red):
        return dict(PT_U=red.PT_U, basis_rom=red.basis_rom, F=table(red))

    need = dict(mass=rom.mdeim_Mh, stiffness=rom.mdeim_Ah, convection=rom.mdeim_Ch, nonlinear_lifting=rom.mdeim_Nh_hat,
                trilinear=rom.mdeim_Nh, lifting=rom.deim_fgh)
    missing = [k for k, v in need.items() if not v]
    if missing:
        raise ValueError(f"operators without a hyper-reductor: {missing}")
    V = rom.basis
    red = rom.mdeim_Nh
    c0 = red._local_values(mus[0], ts[0], u_n=np.zeros(V.shape[0]))  # constant entries (Dirichlet rows)
    W = np.stack([red._local_values(mus[0], ts[0], u_n=V[:, k]) - c0 for k in range(V.shape[1])], axis=1)
    probe = np.random.RandomState(0).standard_normal(V.shape[1])
    again = red._local_values(mus[-1], ts[-1], u_n=V @ probe)
    if not np.allclose(again, W @ probe + c0, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(W).max())):
        raise NotImplementedError("state-dependent operator is not a (mu, t)-independent affine map of the state")
    nl = dict(PT_U=red.PT_U, basis_rom=red.basis_rom, W=W, C=np.broadcast_to(c0, (nt, len(mus), c0.size)).copy())
    return dict(mass=term(need["mass"]), lin=[term(need[k]) for k in ("stiffness", "convection", "nonlinear_lifting")],
                nl=nl, rhs=[term(need["lifting"])], dt=dt, bdf2=(fom.BDF_SCHEME == "2"))


def hrom_terms_on_device(rom, mus):
    """The same term dictionaries as :func:`hrom_terms_from_rom`, with every table of local operator entries built ON THE
    DEVICE by the closed-form P1 assembly (``rt_p1_local_assembly``) instead of ``nt x n_mu`` host calls of the FOM's
    entry-wise assembly per operator (deim.py:429-433 -> fom/base.py:523-599).  The FOM only supplies the scalar functions
    of (mu, t) the closed forms depend on - ``fom.p1_closed_form(mus, ts)``: cell size, diffusivity, lifting amplitudes -
    evaluated vectorised on the host (n_mu calls).  1-D P1 problems on a uniformly scaled mesh (the reference's piston and
    heat problems, fom/nonlinear.py:374-494)."""
    fom = rom.fom
    if not hasattr(fom, "p1_closed_form"):
        raise NotImplementedError("the FOM does not expose closed-form P1 scalars (p1_closed_form)")
    nt, dt = fom.domain["nt"], fom.dt
    ts = dt * np.arange(1, nt + 1)
    cf = fom.p1_closed_form(mus, ts)
    nx, n_mu = int(cf["nx"]), len(mus)
    need = dict(mass=rom.mdeim_Mh, stiffness=rom.mdeim_Ah, convection=rom.mdeim_Ch, nonlinear_lifting=rom.mdeim_Nh_hat,
                trilinear=rom.mdeim_Nh, lifting=rom.deim_fgh)
    missing = [k for k, v in need.items() if not v]
    if missing:
        raise ValueError(f"operators without a hyper-reductor: {missing}")
    flat = lambda a: ops.to_device(np.ascontiguousarray(a, dtype=np.float64).reshape(-1))
    h = flat(cf["h"])

    def entries(red, matrix=True):
        d = np.asarray(red.dofs, dtype=np.int64)
        return (ops.to_device_index(d[:, 0]), ops.to_device_index(d[:, 1]) if matrix else None)

    def table(red, kind, matrix=True, **kw):
        rows, cols = entries(red, matrix)
        return ops.p1_local_assembly(kind, nx, rows, cols, h, **kw).view(nt, n_mu, -1)

    def term(red, F):
        return dict(PT_U=red.PT_U, basis_rom=red.basis_rom, F=F)

    mass = term(need["mass"], table(need["mass"], "mass"))
    lin = [term(need["stiffness"], table(need["stiffness"], "stiffness", coef=flat(cf["alpha"]))),
           term(need["convection"], table(need["convection"], "convection")),
           term(need["nonlinear_lifting"], table(need["nonlinear_lifting"], "trilinear", ramp=flat(cf["lift"])))]
    rhs = [term(need["lifting"], table(need["lifting"], "load", matrix=False, ramp=flat(cf["lift_dot"])))]
    # state-dependent operator: entries are affine in the state, W u_N + C, independent of (mu, t) for the P1 trilinear form
    red = need["trilinear"]
    rows, cols = entries(red)
    V = np.ascontiguousarray(rom.basis.T)                                    # r x N_h: the basis functions as states
    one = ops.to_device(np.ones(V.shape[0] + 1))
    both = ops.p1_local_assembly("trilinear", nx, rows, cols, one, state=np.vstack([V, np.zeros((1, V.shape[1]))]))
    c0 = both[-1]
    W = (both[:-1] - c0[None, :]).T.contiguous()                              # m x r
    nl = dict(PT_U=red.PT_U, basis_rom=red.basis_rom, W=W, C=c0.expand(nt, n_mu, c0.numel()).contiguous())
    return dict(mass=mass, lin=lin, nl=nl, rhs=rhs, dt=dt, bdf2=(fom.BDF_SCHEME == "2"))
