"""SURVEY.md section 8 row f1 - the real caller.  The piston workflow of the reference's driver
(``HyperReducedPiston``, rom/hrom.py:979-1182 with the base class's ``run_offline_rom`` :308-342,
``run_offline_hyperreduction`` :419-452/:1088-1140, ``project_reductors`` :265-272, ``_evaluate`` :504-626): an S-ROM /
ROM pair of ``RomConstructorNonlinear`` with all six operators hyper-reduced, validated against the FOM and certified
by ``compute_rom_difference``.  The fixture tests/golden/hrom.npz holds what the reference's driver + the reference's
classes produce on the closed-form Burgers mock (tests/golden/make_golden.py::gen_hrom).

Two checks:
  * ``PistonWorkflow`` below replays the driver's call sequence on romtime_amd's classes - runs anywhere, on the CPU
    stub and on the HIP path (-m gpu; the reference is not on the GPU box);
  * where /root/reference exists (the build container), the REFERENCE'S OWN DRIVER CODE is run with the class names it
    imported re-pointed to romtime_amd's classes - the drop-in claim itself (host logic on the CPU stub).
Bars: kept-mode counts, interpolation entries, sampled online parameters identical; FOM-space trajectories within 2e-6
rel-L2 (the reference's GMRES stops at 1e-10, rom.py:36 - SURVEY hard part E); error / estimator curves rtol 1e-4."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from romtime_amd.testing import walk_inputs as wi

KEYS = ("a0", "omega", "delta", "alpha_0")


class PistonWorkflow:
    """The call sequence of HyperReducedPiston on romtime_amd's classes (each step cites the driver line it replays)."""

    def __init__(self):
        from romtime_amd import (DiscreteEmpiricalInterpolation, MatrixDiscreteEmpiricalInterpolation,
                                 MatrixDiscreteEmpiricalInterpolationNonlinear, RomConstructorNonlinear)
        from romtime_amd.base import Reductor
        from romtime_amd.conventions import OperatorType as OT

        self.grid = wi.piston_grid()
        rnd = self.rnd = np.random.RandomState(0)
        fom = self.fom = wi.rb_fom()
        fom.exact_solution = None
        walk = {"ts": wi.PISTON_TS, "num_snapshots": None}
        # hrom.py:1003-1038 setup
        self.rom = RomConstructorNonlinear(fom=fom, grid=self.grid, name="ROM")
        self.rom.setup(rnd=rnd)
        self.srom = RomConstructorNonlinear(fom=fom, grid=self.grid, name="S-ROM")
        self.srom.setup(rnd=rnd)
        # hrom.py:274-306, 1040-1086 setup_hyperreduction
        mk = lambda cls, name, fn: cls(name=name, assemble=fn, grid=self.grid, tree_walk_params=dict(walk))
        self.deim_rhs = mk(DiscreteEmpiricalInterpolation, "RHS", fom.assemble_lifting)
        self.mdeim_mass = mk(MatrixDiscreteEmpiricalInterpolation, "Mass", fom.assemble_mass)
        self.mdeim_stiffness = mk(MatrixDiscreteEmpiricalInterpolation, "Stiffness", fom.assemble_stiffness)
        self.deim_rhs.setup(rnd=rnd)
        self.mdeim_mass.setup(rnd=rnd)
        self.mdeim_stiffness.setup(rnd=rnd)
        self.mdeim_convection = mk(MatrixDiscreteEmpiricalInterpolation, OT.CONVECTION, fom.assemble_convection)
        self.mdeim_trilinear_lifting = mk(MatrixDiscreteEmpiricalInterpolation, OT.NONLINEAR_LIFTING,
                                          fom.assemble_nonlinear_lifting)
        self.mdeim_trilinear = mk(MatrixDiscreteEmpiricalInterpolationNonlinear, OT.TRILINEAR, fom.assemble_trilinear)
        self.mdeim_convection.setup(rnd=rnd)
        self.mdeim_trilinear_lifting.setup(rnd=rnd)
        Reductor.setup(self.mdeim_trilinear, rnd=rnd)
        x = np.linspace(0.0, 1.0, fom.Nh)
        self.mdeim_trilinear.rows, self.mdeim_trilinear.cols = self.mdeim_trilinear.get_matrix_topology(
            mu=wi.PISTON_MUS[0], t=1.0, u_n=x)
        self.errors, self.captured = {}, {}

    def run_offline_rom(self, mu_space):  # hrom.py:308-342
        from romtime_amd.conventions import RomParameters as RP

        self.validation_solutions = self.srom.build_reduced_basis(
            num_snapshots=None, mu_space=mu_space, num_basis=None,
            tolerances={RP.TOL_TIME: wi.PISTON_TOL_TIME, RP.TOL_MU: wi.PISTON_TOL_MU})
        self.rom = self.srom.truncate(n=wi.PISTON_SROM_TRUNCATE)
        self.rom.name = "ROM"

    def run_offline_hyperreduction(self, mu_space):  # hrom.py:419-452 then 1088-1140
        from romtime_amd.conventions import OperatorType as OT

        def run(obj, which):                      # _run_deim, hrom.py:819-857
            obj.run(mu_space=mu_space)
            obj.dump_fom_basis()
            for rom in (self.rom, self.srom):
                rom.add_hyper_reductor(reductor=obj, which=which)

        run(self.mdeim_stiffness, OT.STIFFNESS)
        run(self.mdeim_mass, OT.MASS)
        run(self.deim_rhs, OT.RHS)
        run(self.mdeim_convection, OT.CONVECTION)
        run(self.mdeim_trilinear_lifting, OT.NONLINEAR_LIFTING)
        self.mdeim_trilinear.load_fom_basis(basis=self.srom.basis_nonlinear)   # _run_mdeim_nonlinear, hrom.py:1177-1182
        for rom in (self.rom, self.srom):
            rom.add_hyper_reductor(reductor=self.mdeim_trilinear, which=OT.TRILINEAR)

    def project_reductors(self):  # hrom.py:265-272
        self.rom.project_reductors()
        self.srom.project_reductors()

    def evaluate(self, which, mu_space):  # _evaluate, hrom.py:504-626
        from romtime_amd.conventions import Errors
        from romtime_amd.utils import compute_rom_difference

        rom, srom, fom = self.rom, self.srom, self.fom
        out = {}
        for mu in list(mu_space):
            idx = rom.solve(mu=mu, step=which)
            srom.solve(mu=mu, step=which)
            rom.solutions.to_pickle(f"solutions_rom_{rom.N}_{which}_{idx}")
            srom.solutions.to_pickle(f"solutions_srom_{srom.N}_{which}_{idx}")
            if which == "validation":
                uh_fom = self.validation_solutions[idx]
            else:
                fom.setup()
                fom.update_parametrization(mu)
                fom.solve()
                uh_fom = fom.solutions.fom
            uh_rom, uh_srom = rom.solutions.fom, srom.solutions.fom
            nt = uh_fom.shape[1]
            err = lambda uh: np.array([rom._compute_error(uh_fom[:, i], uh[:, i]) for i in range(nt)])
            est = np.array([compute_rom_difference(uN=rom.solutions.rom[:, i], uN_srom=srom.solutions.rom[:, i],
                                                   V_srom=srom.basis) for i in range(rom.solutions.rom.shape[1])])
            out[idx] = {Errors.ESTIMATOR: est, Errors.ROM: err(uh_rom), Errors.SACRIFICIAL: err(uh_srom)}
            self.captured[("rom", which, idx)] = (rom.solutions.rom.copy(), uh_rom.copy())
            self.captured[("srom", which, idx)] = (srom.solutions.rom.copy(), uh_srom.copy())
        self.errors[which] = out

    # ---- f2: artefacts and the resume flow (hrom.py:137-177 dump_*, :344-417 start_from_existing_basis) ------------
    def dump_artefacts(self):
        from romtime_amd.artefacts import dump_offline

        dump_offline(self.rom, self.srom, self.validation_solutions)

    def start_from_existing_basis(self):
        from romtime_amd.artefacts import start_from_existing_basis

        reductors = {k: getattr(self, k) for k in ("deim_rhs", "mdeim_mass", "mdeim_stiffness", "mdeim_convection",
                                                   "mdeim_trilinear_lifting", "mdeim_trilinear")}
        self.rom, self.validation_solutions = start_from_existing_basis(
            self.srom, reductors, {"srom_truncate": wi.PISTON_SROM_TRUNCATE, "srom_num": None, "mdeim_truncate": None})

    def run(self):
        mus = [dict(m) for m in wi.PISTON_MUS]
        self.run_offline_rom(mus)
        self.run_offline_hyperreduction(mus)
        self.project_reductors()
        self.evaluate("validation", self.rom.mu_space["offline"])                       # hrom.py:476-481
        self.evaluate("online", self.rom.build_sampling_space(num=2, rnd=np.random.RandomState(1)))  # hrom.py:483-502
        return self


def _flat_dofs(dofs):
    return np.array(dofs, dtype=np.int64)


def compare_with_reference_run(g, N_rom, N_srom, reductors, srom, online_mus, errors, captured, validation_1):
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert (N_rom, N_srom) == (int(g["N_rom"]), int(g["N_srom"]))
    for name, red in reductors.items():
        assert red.N == int(g[f"N__{name}"]), name
    off = srom.report["offline"]
    assert [off["basis-shape-time"][i] for i in range(3)] == [int(g[f"srom_basis_time_{i}"]) for i in range(3)]
    assert_allclose(off["spectrum-mu"], g["srom_spectrum_mu"], rtol=0, atol=1e-8 * g["srom_spectrum_mu"][0])
    assert_allclose(off["N-spectrum-mu"], g["srom_N_spectrum_mu"], rtol=0, atol=1e-8 * g["srom_N_spectrum_mu"][0])
    # S-ROM basis: same span (its leading N_rom columns are the ROM's basis: same span too)
    for k in (N_rom, N_srom):
        A, B = srom.basis[:, :k], g["srom_basis"][:, :k]
        assert np.linalg.norm(A - B @ (B.T @ A), 2) < 1e-7, k
    # the Mach-number sampler (rom.py:760-860) draws the same online parameters from the same random state
    got = np.array([[m[k] for k in KEYS] for m in online_mus])
    assert_allclose(got, g["online_mus"], rtol=1e-14)
    assert_allclose([m["piston_mach"] for m in online_mus], g["online_mach"], rtol=1e-14)
    assert_allclose(validation_1, g["validation_solution_1"], rtol=0, atol=1e-12)
    # Interpolation entries.  A reductor whose kept singular values are all distinct has its basis columns, hence its
    # greedy entries, pinned by the data: identical to the reference's.  The separable operators (mass, stiffness,
    # nonlinear lifting: two modes with sigma = sqrt(3) twice) leave the rotation inside the cluster to LAPACK; their
    # interpolant is exact on the whole family whatever entries are picked, so nothing downstream depends on it.
    decisive = [name for name in reductors if not bool(g[f"cluster__{name}"])]
    assert "trilinear" in decisive and "rhs" in decisive, decisive
    same_entries = all(np.array_equal(_flat_dofs(reductors[name].dofs), g[f"dofs__{name}"]) for name in decisive)
    worst = 0.0
    for which, n in (("validation", 3), ("online", 2)):
        for idx in range(n):
            for label in ("rom", "srom"):
                uN, uh = captured[(label, which, idx)]
                ref_uh = g[f"{label}_uh__{which}__{idx}"]
                assert uh.shape == ref_uh.shape and uN.shape == g[f"{label}_uN__{which}__{idx}"].shape
                d = rel(uh, ref_uh)
                worst = max(worst, d)
                assert d <= (2e-6 if same_entries else 2e-3), (which, idx, label, d, same_entries)
            for kind, arr in errors[which][idx].items():
                ref = g[f"errors__{which}__{idx}__{kind}"]
                assert_allclose(arr, ref, rtol=1e-4 if same_entries else 5e-2, atol=1e-6 * ref.max(), err_msg=f"{which} {idx} {kind}")
    # the certification the driver exists for: the S-ROM estimator follows the ROM error (same order of magnitude)
    for which, n in (("validation", 3), ("online", 2)):
        for idx in range(n):
            e = errors[which][idx]
            assert 0.3 < e["estimator"].max() / e["rom"].max() < 3.0
    return same_entries, worst


def check_piston_workflow(g, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    w = PistonWorkflow().run()
    reductors = dict(rhs=w.deim_rhs, mass=w.mdeim_mass, stiffness=w.mdeim_stiffness, convection=w.mdeim_convection,
                     nonlinear_lifting=w.mdeim_trilinear_lifting, trilinear=w.mdeim_trilinear)
    same, worst = compare_with_reference_run(g, w.rom.N, w.srom.N, reductors, w.srom, w.rom.mu_space["online"], w.errors,
                                             w.captured, w.validation_solutions[1])
    assert same, "interpolation entries of a reductor with a cluster-free spectrum differ from the reference's"
    # the reductors with a degenerate spectrum: exact on their family at an unseen (mu, t), whatever the entries
    from oracle import romtime_oracle as oracle

    mu_new = dict(a0=12.0, omega=11.0, delta=0.33, alpha_0=0.07)
    for name, fn in (("mass", w.fom.assemble_mass), ("stiffness", w.fom.assemble_stiffness),
                     ("nonlinear_lifting", w.fom.assemble_nonlinear_lifting), ("convection", w.fom.assemble_convection)):
        exact = oracle.eliminate_zeros(fn(mu=mu_new, t=0.31)).data
        got = reductors[name].interpolate(mu=mu_new, t=0.31, which="fom").data
        assert_allclose(got[1:], exact[1:], rtol=0, atol=1e-9 * np.abs(exact).max(), err_msg=name)
    # artefacts the driver leaves behind (hrom.py:530-531; deim.py:166-173 via _run_deim)
    files = set(os.listdir(tmp_path))
    for name in ("basis_fom_mdeim_mass.pkl", "basis_fom_mdeim_stiffness.pkl", "basis_fom_deim_rhs.pkl",
                 "basis_fom_mdeim_convection.pkl", "basis_fom_mdeim_nonlinear-lifting.pkl",
                 f"solutions_rom_{w.rom.N}_validation_0.pkl", f"solutions_srom_{w.srom.N}_online_1.pkl"):
        assert name in files, (name, sorted(files))
    # f2: dump, then resume in a fresh workflow from what is on disk; the resumed pair reproduces the validation errors
    w.dump_artefacts()
    files = set(os.listdir(tmp_path))
    assert set(n for n in g["files_written"] if n.startswith(("basis_", "mu_space", "validation_solutions", "solutions_"))) <= files
    import json

    assert json.load(open("mu_space.json"))["offline"] == [dict(m) for m in wi.PISTON_MUS]
    w2 = PistonWorkflow()
    w2.start_from_existing_basis()
    assert (w2.rom.N, w2.srom.N) == (int(g["resume_N_rom"]), int(g["resume_N_srom"]))
    assert w2.rom.mu_space["online"] == [] and w2.rom.mu_space["offline"] == [dict(m) for m in wi.PISTON_MUS]
    w2.project_reductors()
    w2.evaluate("validation", w2.rom.mu_space["offline"])
    for idx in range(3):
        for kind, arr in w2.errors["validation"][idx].items():
            ref = g[f"resume_errors__{idx}__{kind}"]
            assert_allclose(arr, ref, rtol=1e-4, atol=1e-6 * ref.max(), err_msg=f"resume {idx} {kind}")
            assert_allclose(arr, w.errors["validation"][idx][kind], rtol=1e-9, atol=1e-12)
    print(f"piston workflow: FOM-space trajectories within {worst:.2e} rel-L2 of the reference run")
    try:
        import json

        rep = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "hrom_flow_report.jsonl")
        os.makedirs(os.path.dirname(rep), exist_ok=True)
        with open(rep, "a") as fp:
            fp.write(json.dumps(dict(test="piston workflow vs the reference driver's run", worst_rel_l2_fom_space=worst,
                                     N_rom=int(w.rom.N), N_srom=int(w.srom.N))) + "\n")
    except OSError:
        pass


@pytest.fixture(scope="module")
def golden_hrom():
    from tests.conftest import load_golden

    return load_golden("hrom.npz")


def test_piston_workflow_hostlogic(cpu_ops, golden_hrom, tmp_path, monkeypatch):
    check_piston_workflow(golden_hrom, tmp_path, monkeypatch)


@pytest.mark.gpu
def test_piston_workflow_hip(golden_hrom, tmp_path, monkeypatch):
    check_piston_workflow(golden_hrom, tmp_path, monkeypatch)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference tree (build container only)")
def test_reference_driver_over_romtime_amd_classes(cpu_ops, golden_hrom):
    """The drop-in claim: rom/hrom.py's OWN code (HyperReducedPiston.run_offline_rom, run_offline_hyperreduction,
    project_reductors, evaluate_validation, evaluate_online) with ``RomConstructorNonlinear``, the three (M)DEIM
    classes and ``compute_rom_difference`` re-pointed to romtime_amd's reproduces the run it makes on its own classes."""
    import romtime_amd
    from romtime_amd import utils
    from romtime_amd.base import Reductor
    from tests.golden import make_golden as mg

    ref = mg.load_reference()
    classes = dict(rom=romtime_amd.RomConstructorNonlinear, deim=romtime_amd.DiscreteEmpiricalInterpolation,
                   mdeim=romtime_amd.MatrixDiscreteEmpiricalInterpolation,
                   nmdeim=romtime_amd.MatrixDiscreteEmpiricalInterpolationNonlinear, reductor=Reductor,
                   compute_rom_difference=utils.compute_rom_difference)
    data, H = mg.run_piston_driver(ref, classes)
    assert type(H.rom).__module__.startswith("romtime_amd") and type(H.mdeim_trilinear).__module__.startswith("romtime_amd")
    g = golden_hrom
    for key in ("N_rom", "N_srom", "N__rhs", "N__mass", "N__stiffness", "N__convection", "N__nonlinear_lifting",
                "N__trilinear"):
        assert int(data[key]) == int(g[key]), key
    assert_allclose(data["online_mus"], g["online_mus"], rtol=1e-14)
    assert list(data["files_written"]) == list(g["files_written"])              # same artefacts, same names
    for key in g.files:
        if key.startswith("dofs__") and not bool(g["cluster__" + key[6:]]):
            np.testing.assert_array_equal(data[key], g[key], err_msg=key)
        elif key.startswith(("rom_uh__", "srom_uh__")):
            assert np.linalg.norm(data[key] - g[key]) <= 2e-6 * np.linalg.norm(g[key]), key
        elif key.startswith(("errors__", "resume_errors__")):   # resume_*: start_from_existing_basis (hrom.py:344-417)
            assert_allclose(data[key], g[key], rtol=1e-4, atol=1e-6 * g[key].max(), err_msg=key)
    assert (int(data["resume_N_rom"]), int(data["resume_N_srom"])) == (int(g["resume_N_rom"]), int(g["resume_N_srom"]))
