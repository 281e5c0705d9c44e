"""CPU-side checks of the drop-in boundary: the HIP library loads without a GPU, exports
every symbol ``include/gvl_hip.h`` declares, the ctypes mirrors of the C structs have the
C layout, and argument validation happens before anything touches a device.
No compute is launched here."""

import ctypes as C
import re
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
HEADER = REPO / "include" / "gvl_hip.h"


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build_hip()
    from genvarloader_amd import _lib

    return _lib.load()


def declared_symbols():
    txt = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(gvl_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported(lib):
    from genvarloader_amd import _lib

    decl = declared_symbols()
    assert decl, "no declarations parsed"
    assert sorted(_lib.SYMBOLS) == decl
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in gvl_hip.h but not exported"
    assert lib.gvl_abi_version() == _lib.ABI_VERSION == 4


def test_ctypes_structs_match_c_layout(tmp_path):
    from genvarloader_amd import _lib

    structs = {"gvl_static": _lib.GvlStatic, "gvl_batch": _lib.GvlBatch, "gvl_out": _lib.GvlOut,
               "gvl_loader_config": _lib.GvlLoaderConfig, "gvl_loader_batch": _lib.GvlLoaderBatch}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, st in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('printf("gvl_vrec %zu\\n", sizeof(gvl_vrec));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True)
               .stdout.strip().splitlines())
    for cname, st in structs.items():
        assert int(got[cname]) == C.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"
    assert int(got["gvl_vrec"]) == 16


def test_argument_validation_without_device(lib):
    from genvarloader_amd import _lib

    st, bt, out = _lib.GvlStatic(), _lib.GvlBatch(), _lib.GvlOut()
    assert lib.gvl_reconstruct(None, None, None, None) == 1
    assert b"NULL" in lib.gvl_last_error()
    bt.batch, bt.ploidy = 4, 0
    assert lib.gvl_reconstruct(C.byref(st), C.byref(bt), C.byref(out), None) == 1
    bt.ploidy = 2
    assert lib.gvl_reconstruct(C.byref(st), C.byref(bt), C.byref(out), None) == 1  # no output buffer
    assert b"output" in lib.gvl_last_error()
    assert lib.gvl_onehot(None, C.c_int64(-1), None, None) == 1
    assert lib.gvl_onehot(None, C.c_int64(0), None, None) == 0
    assert lib.gvl_rc_rows(None, None, None, C.c_int64(0), None) == 0
    assert lib.gvl_rc_rows(None, None, None, C.c_int64(3), None) == 1
    with pytest.raises(ValueError):
        _lib.check(1)


def test_no_cpu_fallback_and_oracle_isolation():
    """The product package must not import the oracle, and must fail loudly when the HIP
    library is missing."""
    pkg = REPO / "genvarloader_amd"
    for f in pkg.rglob("*.py"):
        txt = f.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, f
    code = ("import os; os.environ['GVL_HIP_LIB']='/nonexistent/libgvl_hip.so';"
            "from genvarloader_amd import _lib\n"
            "try:\n _lib.load(); print('LOADED')\n"
            "except _lib.GvlError as e: print('RAISED')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=REPO)
    assert r.stdout.strip() == "RAISED", r.stdout + r.stderr
