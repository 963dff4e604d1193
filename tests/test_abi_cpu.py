"""CPU-side checks of the drop-in boundary: the library loads and exports what the header declares."""
import os
import re

from romtime_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "romtime_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/romtime_hip.h but not exported"
    assert lib.rt_version() >= 100


def test_binding_table_matches_header():
    assert set(_lib.SIGNATURES) == _declared()


def test_header_cites_reference_for_each_hot_entry_point():
    text = open(os.path.join(REPO, "include", "romtime_hip.h")).read()
    for needle in ("pod.py", "deim.py:517-561", "utils.py:96-113", "mdeim.py:153-192", "rom.py"):
        assert needle in text


def test_package_shutdown_is_harmless_without_a_gpu_and_twice():
    """romtime_amd.shutdown() releases device state in order; with nothing created (no GPU here) it is a no-op, also
    when called again, and the hot path still refuses to run without a GPU afterwards."""
    import numpy as np
    import pytest
    import torch

    import romtime_amd
    from romtime_amd._lib import Context, RomtimeHipError

    romtime_amd.shutdown()
    romtime_amd.shutdown()
    assert len(Context._live) == 0
    if not torch.cuda.is_available():
        with pytest.raises(RomtimeHipError):
            romtime_amd.orth(np.ones((8, 3)))


def test_header_documents_every_context_option_and_counter():
    """Every name rt_ctx_set_option / rt_ctx_get_counter accepts (csrc/api.hip) appears in include/romtime_hip.h."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "romtime_amd", "csrc", "api.hip")).read()
    header = open(os.path.join(root, "include", "romtime_hip.h")).read()
    options = set(re.findall(r'key == "([a-z_0-9]+)"', src))
    counters = set(re.findall(r'\{"([a-z_0-9]+)", RT_CNT_', src))
    assert {"gram_pace", "cu_limit", "eig_xcd", "eig_one_xcd", "sweep_graph"} <= options
    missing = [name for name in sorted(options | counters) if f'"{name}"' not in header]
    assert not missing, f"undocumented in include/romtime_hip.h: {missing}"


def test_gram_plan_choice_is_host_logic_and_matches_the_measured_rules():
    """rt_gram_plan_info: the form rt_gram takes (no GPU needed).  One paced launch with uniform slots where the slot model
    puts it within 5 % of two launches (profiles/r03_gram_pace_ab.txt), two launches elsewhere, the
    generic GEMM outside the kernel's range."""
    import ctypes as C

    from romtime_amd import _lib

    lib = _lib.load()

    def plan(cus, rows, cols):
        out = (C.c_int * 5)()
        assert lib.rt_gram_plan_info(cus, rows, cols, out) == 0
        return tuple(out)

    assert plan(256, 1_000_000, 512)[:3] == (2, 7, 5)            # bench shape on the whole chip: 62 of 64 slots
    assert plan(256, 1_000_000, 500)[:3] == (2, 7, 5)
    assert plan(256, 1_000_000, 384)[0] == 2 and plan(256, 1_000_000, 256)[0] == 2 and plan(256, 400_000, 1024)[:3] == (2, 2, 1)
    assert plan(256, 600_000, 640)[0] == 1 and plan(256, 500_000, 768)[0] == 1     # model 1.15 / 1.07: two launches
    assert plan(224, 1_000_000, 512) == (1, 6, 4, 9, 14)          # the POD pipeline's 224-CU stream: 56 slots, two launches
    assert plan(256, 100_000, 256) == (1, 16, 10, 16, 16)         # two tile columns, cap binding: two launches (C2)
    assert plan(256, 200_000, 256)[0] == 2 and plan(256, 50_000, 512)[:3] == (2, 7, 5) and plan(256, 30_000, 384)[0] == 2
    assert plan(256, 1_000_000, 96)[0] == 0 and plan(256, 4000, 512)[0] == 0 and plan(256, 1_000_000, 1100)[0] == 0
    assert lib.rt_gram_plan_info(256, 0, 512, (C.c_int * 5)()) < 0
