"""The C-ABI shared library loads on a CPU-only box and exports exactly what
include/spx.h declares (no compute calls here)."""
import os
import re

import pytest

from spearmint_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    return os.path.exists(engine.default_lib_path())


@pytest.fixture(scope="module")
def lib():
    if not _built():
        import __graft_entry__ as g
        g.build()
    return engine.load_library()


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "spx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spx_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    syms = _header_symbols()
    assert len(syms) >= 20
    assert sorted(engine.ABI) == syms


def test_every_symbol_exported(lib):
    for name in _header_symbols():
        assert hasattr(lib, name), name


def test_version_and_error_string(lib):
    assert lib.spx_version() >= 100
    assert isinstance(lib.spx_last_error(), bytes)
    assert lib.spx_timing_name(0) == b"scale_rows"


def test_create_is_lazy_and_arg_checks(lib):
    # spx_create must not touch the GPU (the chooser is constructed before a fork)
    eng = engine.Engine(0)
    with pytest.raises(ValueError):
        eng.set_hypers([[0.0, 1e-3, 1.0, 1.0]])      # observations not set yet
    with pytest.raises(ValueError):
        eng.ei_run()
    eng.close()


def test_no_cpu_fallback(lib):
    """Without a HIP device the product path fails loudly instead of computing on the CPU."""
    if engine.device_count() > 0:
        pytest.skip("a GPU is present")
    eng = engine.Engine(0)
    with pytest.raises(engine.SpxError):
        eng.ei_grid([[0.1, 0.2], [0.3, 0.4]], [1.0, 2.0], [[0.5, 0.5]], [[0.0, 1e-3, 1.0, 1.0, 1.0]])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "spearmint_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|import_module\(.oracle|oracle/", txt, re.M), \
                    os.path.join(d, f)
