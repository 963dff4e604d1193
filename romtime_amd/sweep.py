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
