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
    assert lib.gvl_abi_version() == _lib.ABI_VERSION == 11


def test_header_enums_match_the_python_mirror(lib):
    """The constants of ``include/gvl_hip.h`` that ``_lib.py`` restates: tuning keys (and that the library rejects the first key
    past them), the loader's table / slot parts, the one-hot layouts."""
    from genvarloader_amd import _lib

    txt = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    enums = {k: int(v) for k, v in re.findall(r"\b(GVL_[A-Z0-9_]+)\s*=\s*(-?\d+)", txt)}
    defines = {k: int(v) for k, v in re.findall(r"#define\s+(GVL_[A-Z0-9_]+)\s+(-?\d+)\b", txt)}
    both = {**defines, **enums}
    for name in ("PIPE_ROWS_X100", "PIPE_MIN_ROWS", "LEAN_SUB", "TRACK_PLAN_MAX_MB", "RAGGED_SIZING", "HAP_PLAN_MAX_MB", "MIXED_MIN_ROWS"):
        assert both["GVL_TUNE_" + name] == getattr(_lib, "TUNE_" + name), name
    n_keys = both["GVL_TUNE_COUNT"]
    assert n_keys == 7
    assert lib.gvl_set_tuning(n_keys - 1, 0) == 0 and lib.gvl_set_tuning(n_keys, 0) != 0 and lib.gvl_set_tuning(-1, 0) != 0
    assert both["GVL_LOADER_TABLE_PARTS"] == _lib.LOADER_TABLE_PARTS
    assert both["GVL_ONEHOT_LC"] == _lib.GVL_ONEHOT_LC and both["GVL_ONEHOT_CL"] == _lib.GVL_ONEHOT_CL
    assert both["GVL_ABI_VERSION"] == _lib.ABI_VERSION


def test_ctypes_structs_match_c_layout(tmp_path):
    from genvarloader_amd import _lib

    structs = {"gvl_static": _lib.GvlStatic, "gvl_batch": _lib.GvlBatch, "gvl_out": _lib.GvlOut,
               "gvl_loader_config": _lib.GvlLoaderConfig, "gvl_loader_batch": _lib.GvlLoaderBatch,
               "gvl_track_set": _lib.GvlTrackSet, "gvl_ref_batch": _lib.GvlRefBatch, "gvl_svar2_batch": _lib.GvlSvar2Batch}
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
    # the SVAR2 provider: sizes are host arithmetic, arguments are checked before the device is touched
    assert lib.gvl_svar2_workspace_bytes(4, 2, 10, 64, 100) > 0 and lib.gvl_svar2_workspace_bytes(4, 0, 10, 64, 100) == 0
    assert lib.gvl_svar2_workspace_bytes(4, 2, 10, 64, 100) % 256 == 0
    sv, merged, goi = _lib.GvlSvar2Batch(), _lib.GvlStatic(), C.c_void_p()
    assert lib.gvl_svar2_merge(None, None, None, C.c_int64(3), C.c_int64(0), C.c_int64(1), None, C.c_int64(0), None, None, None) == 1
    assert lib.gvl_svar2_merge(C.byref(st), C.byref(sv), None, C.c_int64(3), C.c_int64(2), C.c_int64(2), None, C.c_int64(0),
                               C.byref(merged), C.byref(goi), None) == 1
    assert b"workspace" in lib.gvl_last_error()
    bt2 = _lib.GvlBatch(batch=1, ploidy=1, out_offsets=8, out_bounds=8, output_length=-1)
    out2 = _lib.GvlOut(haps=8)
    assert lib.gvl_reconstruct(C.byref(st), C.byref(bt2), C.byref(out2), None) == 1      # out_offsets and out_bounds are exclusive


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


def test_loader_slot_and_table_layout(lib):
    """gvl_loader_slot_bytes / gvl_loader_table_bytes are pure host arithmetic: the parts of a ring slot for the
    reference loader's output kinds (fixed, annotated, ragged, haplotypes + tracks) and of the epoch table."""
    from genvarloader_amd import _lib

    def slot(**kw):
        cfg = _lib.GvlLoaderConfig(n_regions=10, n_samples=4, ploidy=2, batch_size=8, **kw)
        parts = (C.c_int64 * _lib.LOADER_SLOT_PARTS)()
        lib.gvl_loader_slot_bytes.restype = C.c_int64
        return int(lib.gvl_loader_slot_bytes(C.byref(cfg), parts)), [int(x) for x in parts]

    up = lambda x: (x + 255) & ~255
    K, L = 16, 300
    n, p = slot(output_length=L, want_onehot=1, want_haps=1)
    assert p[0] == 0 and p[1] == up(4 * K * L) and n == up(4 * K * L) + up(K * L) + up(8 * (K + 1)) + 256
    n_a, p_a = slot(output_length=L, want_annot=1)                     # annotations imply the haplotype bytes
    assert n_a == up(K * L) + up(8 * (K + 1)) + 2 * up(4 * K * L) + 256 and p_a[8] - p_a[7] == up(4 * K * L)
    n_r, _ = slot(output_length=-1, max_row_len=500, want_haps=1, deterministic=1)
    assert n_r == up(K * 500) + up(8 * (K + 1)) + 256
    assert slot(output_length=-1, want_haps=1)[0] <= 0                 # ragged rows need a bound
    lib.gvl_tracks_scratch_bytes.restype = C.c_int64
    scr = int(lib.gvl_tracks_scratch_bytes(C.c_int64(8), C.c_int64(2), C.c_int64(700)))
    n_t, p_t = slot(output_length=L, want_haps=1, n_tracks=3, scratch_stride=700)
    assert p_t[10] - p_t[9] == up(4 * 3 * K * L) and p_t[11] - p_t[10] == up(scr)
    cfg = _lib.GvlLoaderConfig(ploidy=2, batch_size=8)
    po = (C.c_int64 * _lib.LOADER_TABLE_PARTS)()
    lib.gvl_loader_table_bytes.restype = C.c_int64
    nb = int(lib.gvl_loader_table_bytes(C.byref(cfg), C.c_int64(100), po))
    assert [int(x) for x in po][:5] == [0, up(1600), up(1600) + up(1600), up(1600) + up(1600) + up(800), up(1600) + up(1600) + up(800) + 256]
    assert nb == int(po[4]) + up(8 * 13) and int(po[5]) == int(po[6]) == nb          # no tracks: the two sizing parts are empty
    cfg_t = _lib.GvlLoaderConfig(ploidy=2, batch_size=8, n_tracks=1)
    nb_t = int(lib.gvl_loader_table_bytes(C.byref(cfg_t), C.c_int64(100), po))
    # with tracks: every batch's (batch_size + 1) scratch-track offsets + the batch_size * ploidy + 1 row offsets
    assert int(po[6]) - int(po[5]) == up(8 * (100 + 13)) and nb_t - int(po[6]) == up(8 * 17)
