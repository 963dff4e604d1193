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
