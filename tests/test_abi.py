"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/mellon_hip.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mellon_amd import _build, _lib
    _build.build(verbose=False)
    return _lib.load_library()


def test_header_and_binding_agree(lib):
    from mellon_amd import _lib
    header = open(os.path.join(ROOT, "include", "mellon_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(mln_\w+)\s*\(", header, flags=re.M))
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)


def test_no_cpu_fallback(lib):
    import ctypes as C
    has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        pytest.skip("GPU present")
    from mellon_amd import _lib
    with pytest.raises(_lib.MellonHipError, match="no HIP device|no CPU fallback|failed"):
        _lib.Context(0)
    from mellon_amd import cov
    with pytest.raises(_lib.MellonHipError):
        cov.Matern52(1.0)(np.zeros((3, 2)), np.zeros((2, 2)))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no product file may import, load or execute it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|import_module\(\s*[\"']oracle|oracle[/\\][\w_]+\.(py|so|c)", re.M)
    pkg = os.path.join(ROOT, "mellon_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), f


def test_covariance_lowering_and_json():
    from mellon_amd import cov
    from mellon_amd.base_cov import Covariance
    c = cov.Matern52(1.5, active_dims=slice(None, -1)) * cov.Matern52(0.4, active_dims=-1) + 2.0
    low = c.lower(5)
    assert [tuple(l[3]) for l in low.leaves] == [(0, 1, 2, 3), (4,)]
    assert [t[0] for t in low.toks] == [0, 0, 3, 1, 2]          # LEAF LEAF MUL CONST ADD
    nested = cov.Mul(cov.ExpQuad(1.0, active_dims=[1, 2]), 3.0, active_dims=[4, 0, 2])
    assert tuple(nested.lower(6).leaves[0][3]) == (0, 2)         # composite dims first, then the child's
    state = c.to_dict()
    assert state["type"] == "mellon.Covariance" and state["metadata"]["classname"] == "Add"
    assert state["left_data"]["left_data"]["metadata"]["module_name"] == "mellon.cov"
    c2 = Covariance.from_json(c.to_json())
    assert repr(c2) == repr(c)
    from oracle import mellon_oracle as mo
    oc = mo.Covariance.from_dict(state)                          # the oracle reads the same wire format
    assert isinstance(oc, mo.Add) and oc.right == 2.0


def test_tools_compile():
    """The measurement scripts under tools/ are run by hand on the GPU box; here at least every one of them parses, and the
    shell scripts pass `bash -n`."""
    import subprocess
    tools = os.path.join(ROOT, "tools")
    for f in sorted(os.listdir(tools)):
        path = os.path.join(tools, f)
        if f.endswith(".py"):
            compile(open(path).read(), path, "exec")
        elif f.endswith(".sh"):
            assert subprocess.run(["bash", "-n", path]).returncode == 0, f
