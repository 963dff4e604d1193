"""Solution containers with the reference's shape conventions (src/romtime/base.py:19-79):
``fom`` is N_h x nt, ``rom`` is r x nt, ``ts`` an array of time instants."""
from __future__ import annotations

from copy import deepcopy

import numpy as np


class SolutionsStorage:
    def __init__(self, ts, mu, domain, fom, snapshots=None) -> None:
        self.ts = np.array(ts)
        self.mu = deepcopy(mu)
        self.snapshots = deepcopy(snapshots)
        self.fom = deepcopy(fom)
        self.domain = deepcopy(domain)

    def to_pickle(self, name):
        import pickle

        with open(name + ".pkl", mode="wb") as fp:
            pickle.dump(self, fp)


class RomSolutionsStorage(SolutionsStorage):
    def __init__(self, ts, mu, domain, fom, rom) -> None:
        super().__init__(ts=ts, mu=mu, domain=domain, fom=fom)
        self.rom = deepcopy(rom)
