"""``Reductor``: parameter-space bookkeeping shared by the ROM and (M)DEIM classes.

Same attributes, report keys and method semantics as ``src/romtime/rom/base.py:9-163``
(``mu_space``, ``report``, ``errors_rom``, ``add_mu`` returning ``(idx, mu)``,
``build_sampling_space`` -> sklearn ``ParameterSampler``)."""
from __future__ import annotations

from collections import defaultdict

import numpy as np

from .conventions import ProblemType, Stage, Treewalk, TreewalkNonlinear
from .utils import compute_error


class Reductor:
    """Base of every reduction object.  ``FOM`` / ``ROM`` select the coordinates an interpolant is
    returned in; the ``BASIS_*`` / ``SPECTRUM_*`` / ``ENERGY_*`` attributes are the report keys of
    the tree walk (same strings as ``romtime.conventions.Treewalk``)."""

    FOM, ROM = ProblemType.FOM, ProblemType.ROM

    def __init__(self, grid=None) -> None:
        self.grid = grid
        self.mu_space = {Stage.OFFLINE: [], Stage.ONLINE: [], Stage.VALIDATION: []}
        self.report = defaultdict(dict)
        self.errors_rom = defaultdict(list)
        self.summary_errors = None
        self.mu = None
        self.random_state = None

    @staticmethod
    def _compute_error(u, ue):
        """||u - ue||_2 / sqrt(N) (rom/base.py:52-73)."""
        return compute_error(u, ue)

    def add_mu(self, step, mu):
        """Append ``mu`` to the stage's list; the index is that of its FIRST occurrence
        (``list.index``, rom/base.py:88-90)."""
        self.mu_space[step].append(mu)
        self.mu = mu
        return self.mu_space[step].index(mu), mu

    def build_sampling_space(self, num, rnd=None):
        from sklearn.model_selection import ParameterSampler

        return ParameterSampler(param_distributions=self.grid, n_iter=num, random_state=rnd)

    def setup(self, rnd=None):
        """Reset the offline report skeleton (rom/base.py:122-152)."""
        self.random_state = rnd
        off = self.report[Stage.OFFLINE]
        for names in (Treewalk, TreewalkNonlinear):
            off[names.BASIS_AFTER_WALK] = None
            off[names.BASIS_FINAL] = None
            off[names.SPECTRUM_MU] = None
            off[names.ENERGY_MU] = None
            off[names.BASIS_TIME] = dict()
            off[names.SPECTRUM_TIME] = dict()
            off[names.ENERGY_TIME] = dict()

    def create_errors_summary(self):
        import pandas as pd

        rows = {
            idx: dict(mean=np.mean(err), median=np.median(err), max=np.max(err), min=np.min(err))
            for idx, err in self.errors_rom.items()
        }
        self.summary_errors = pd.DataFrame(rows).T


for _key in Treewalk.keys():  # Reductor.BASIS_FINAL, Reductor.SPECTRUM_TIME, ...
    setattr(Reductor, _key, getattr(Treewalk, _key))
