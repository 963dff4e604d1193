"""String keys shared with callers of the romtime class surface.

The values (not the layout of this file) are the contract: report dictionaries, tree-walk
parameter dictionaries and on-disk artefact names written by romtime's drivers use exactly
these strings (``src/romtime/conventions.py:4-156``), so existing ``setup.json`` /
``mu_space.json`` / ``basis_*.pkl`` artefacts and caller code keep working.
"""
from __future__ import annotations


def _ns(name, doc, **values):
    cls = type(name, (), dict(values))
    cls.__doc__ = doc
    cls.keys = staticmethod(lambda: tuple(values))
    return cls


_PROBLEMS = dict(FOM="fom", ROM="rom", SROM="srom", HROM="hrom")

ProblemType = _ns("ProblemType", "which model a quantity belongs to", **_PROBLEMS)
Errors = _ns("Errors", "keys of the per-parameter error payloads the driver writes (conventions.py:36-44)", **_PROBLEMS,
             SACRIFICIAL="sacrificial", ESTIMATOR="estimator", AVERAGE_ROM="rom_average",
             AVERAGE_ESTIMATOR="estimator_average", AVERAGE_SACRIFICIAL="srom_average")
Stage = _ns("Stage", "phase of the reduction workflow", OFFLINE="offline", VALIDATION="validation", ONLINE="online")
BDF = _ns("BDF", "time scheme order, as the FOM's BDF_SCHEME attribute spells it", ONE="1", TWO="2")
EmpiricalInterpolation = _ns("EmpiricalInterpolation", "hyper-reductor kinds", DEIM="DEIM", MDEIM="MDEIM",
                             NONLINEAR="N-MDEIM")
OperatorType = _ns(
    "OperatorType", "algebraic operators a hyper-reductor can stand for",
    **_PROBLEMS, CONVECTION="convection", FORCING="forcing", LIFTING="lifting", MASS="mass",
    TRILINEAR="trilinear", NONLINEAR="nonlinear", NONLINEAR_LIFTING="nonlinear-lifting",
    REDUCED_BASIS="reduced-basis", RHS="rhs", STIFFNESS="stiffness",
)
StorageNames = _ns(
    "StorageNames", "artefact file names in the working directory",
    ROM="basis_rom.pkl", SROM="basis_srom.pkl", VALIDATION_SOLUTIONS="validation_solutions.pkl",
    SETUP="setup.json", MU_SPACE="mu_space.json", MU_SPACE_DEIM="mu_space_deim.json",
)
RomParameters = _ns(
    "RomParameters", "keys of the tree-walk / driver parameter dictionaries",
    NUM_ONLINE="num_online", SROM_TRUNCATE="srom_truncate", SROM_KEEP="srom_num", NMDEIM_SIZE="mdeim_truncate",
    NUM_BASIS="num_phi", NUM_MU="num_mu", NUM_SNAPSHOTS="num_snapshots", NUM_TIME="num_time",
    TOL_BASIS="tol_phi", TOL_MU="tol_mu", TOL_TIME="tol_time", TS="ts",
)
PistonParameters = _ns("PistonParameters", "piston problem parameter names", ALPHA="alpha", DELTA="delta",
                       GAMMA="gamma", OMEGA="omega", A0="a0", MACH_PISTON="piston_mach", NONLINEARITY="eta")


def _treewalk(prefix):
    stems = dict(BASIS_AFTER_WALK="basis-shape-after-tree-walk", BASIS_FINAL="basis-shape-final",
                 BASIS_TIME="basis-shape-time", ENERGY_MU="energy-mu", ENERGY_TIME="energy-time",
                 SPECTRUM_MU="spectrum-mu", SPECTRUM_TIME="spectrum-time")
    return {k: prefix + v for k, v in stems.items()}


Treewalk = _ns("Treewalk", "report keys of the solution tree walk", **_treewalk(""))
TreewalkNonlinear = _ns("TreewalkNonlinear", "report keys of the nonlinear-term tree walk", **_treewalk("N-"))
