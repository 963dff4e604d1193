"""Solution containers with the reference's shape conventions (src/romtime/base.py:19-79):
``fom`` is N_h x nt, ``rom`` is r x nt, ``ts`` an array of time instants; every field is an
independent copy of what the solver handed in."""
from __future__ import annotations

import copy
import pickle

import numpy as np


class SolutionsStorage:
    _FIELDS = ("mu", "domain", "fom", "snapshots")

    def __init__(self, ts, mu, domain, fom, snapshots=None) -> None:
        given = dict(mu=mu, domain=domain, fom=fom, snapshots=snapshots)
        self.ts = np.array(ts)
        for name in self._FIELDS:
            setattr(self, name, copy.deepcopy(given[name]))

    def to_pickle(self, name):
        with open(f"{name}.pkl", "wb") as fp:
            pickle.dump(self, fp)


class RomSolutionsStorage(SolutionsStorage):
    """Adds the reduced coefficients (r x nt) next to the lifted FOM-space solution."""

    def __init__(self, ts, mu, domain, fom, rom) -> None:
        SolutionsStorage.__init__(self, ts, mu, domain, fom)
        self.rom = copy.deepcopy(rom)
