"""On-disk artefacts of an offline phase, with the reference's file names and payloads, and the resume flow that loads
them straight onto the GPU.

What the reference's driver writes (rom/hrom.py:137-177, conventions.py:4-12, deim.py:77-81,166-173) and reads back in
``start_from_existing_basis`` (rom/hrom.py:344-417):

    mu_space.json                       ROM parameter space {offline, online, validation} (json / ujson)
    basis_rom.pkl, basis_srom.pkl       reduced bases, plain pickled ndarrays (N_h x N)
    basis_fom_{deim|mdeim|n-mdeim}_{name}.pkl   collateral bases (written by each reductor's dump_fom_basis)
    basis_fom_n-mdeim_trilinear.pkl     the nonlinear-term basis of the S-ROM (dump_nonlinear_basis)
    validation_solutions.pkl            {mu_idx: FOM solution} of the offline parameters

Artefacts produced by the reference load here unchanged and vice versa: every payload is a NumPy array, a dict of them
or JSON."""
from __future__ import annotations

from .conventions import OperatorType, RomParameters, Stage, StorageNames
from .utils import dump_json, dump_pickle, read_json, read_pickle

NONLINEAR_BASIS = f"basis_fom_n-mdeim_{OperatorType.TRILINEAR}.pkl"  # hrom.py:166-168

# the operator slot each reductor of the piston workflow is attached to when resuming (hrom.py:389-417)
RESUME_SLOTS = (("deim_rhs", OperatorType.LIFTING), ("mdeim_mass", OperatorType.MASS), ("mdeim_stiffness", OperatorType.STIFFNESS),
                ("mdeim_convection", OperatorType.CONVECTION), ("mdeim_trilinear_lifting", OperatorType.NONLINEAR_LIFTING),
                ("mdeim_trilinear", OperatorType.TRILINEAR))


def dump_offline(rom, srom, validation_solutions=None):
    """dump_mu_space + dump_reduced_basis + dump_nonlinear_basis + dump_validation_fom (hrom.py:137-177); the
    collateral bases are written by the reductors themselves (``dump_fom_basis``, deim.py:166-173)."""
    dump_json(StorageNames.MU_SPACE, rom.mu_space)
    dump_pickle(StorageNames.ROM, rom.basis)
    dump_pickle(StorageNames.SROM, srom.basis)
    if getattr(srom, "basis_nonlinear", None) is not None:
        dump_pickle(NONLINEAR_BASIS, srom.basis_nonlinear)
    if validation_solutions is not None:
        dump_pickle(StorageNames.VALIDATION_SOLUTIONS, validation_solutions)


def start_from_existing_basis(srom, reductors, rom_params):
    """Resume from the artefacts in the working directory (hrom.py:344-417).

    ``srom``: a set-up ``RomConstructor*``; ``reductors``: dict with the keys of ``RESUME_SLOTS`` -> set-up (M)DEIM objects
    (topology known, no basis yet); ``rom_params``: the driver's dictionary (``srom_truncate``; optional ``srom_num`` =
    how many S-ROM modes to keep, ``mdeim_truncate`` = size of the N-MDEIM basis).  Returns ``(rom, validation_solutions)``:
    the truncated ROM with every reductor attached to it and to ``srom`` (project them with ``project_reductors``)."""
    try:
        validation = read_pickle(StorageNames.VALIDATION_SOLUTIONS)
    except FileNotFoundError:
        validation = None
    try:
        mu_space = read_json(StorageNames.MU_SPACE)
    except FileNotFoundError:
        mu_space = {Stage.OFFLINE: [], Stage.ONLINE: [], Stage.VALIDATION: []}
    basis_srom = read_pickle(StorageNames.SROM)
    keep = rom_params.get(RomParameters.SROM_KEEP, None)
    if keep is not None:
        basis_srom = basis_srom[:, :keep]
    srom.load_from_basis(basis=basis_srom, mu_space=mu_space)
    rom = srom.truncate(rom_params[RomParameters.SROM_TRUNCATE])
    for key, which in RESUME_SLOTS:
        red = reductors.get(key)
        if red is None:
            continue
        if key == "mdeim_trilinear":
            red.load_fom_basis(keep=rom_params.get(RomParameters.NMDEIM_SIZE, None))
        else:
            red.load_fom_basis()
        for target in (rom, srom):
            target.add_hyper_reductor(reductor=red, which=which)
    return rom, validation
