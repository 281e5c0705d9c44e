"""Loader for the pickle-free fixtures written by tests/golden/make_fixtures.py."""

from __future__ import annotations

from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def _unbox(x):
    return x[()] if x.ndim == 0 else x


def load_ref_cases(name: str):
    """-> list of (inputs tuple with None for absent entries, expected)."""
    z = np.load(GOLDEN / f"ref_{name}.npz")
    out = []
    for i in range(int(z["n"])):
        n_in = int(z[f"{i}/n_in"])
        inputs = tuple(_unbox(z[f"{i}/in{j}"]) if f"{i}/in{j}" in z.files else None
                       for j in range(n_in))
        if f"{i}/exp" in z.files:
            exp = z[f"{i}/exp"]
        else:
            exp = tuple(z[f"{i}/exp{j}"] for j in range(int(z[f"{i}/n_exp"])))
        out.append((inputs, exp))
    return out


def load_pyref(name: str) -> dict:
    z = np.load(GOLDEN / f"pyref_{name}.npz")
    d = {k: _unbox(z[k]) for k in z.files}
    for k in ("to_rc", "keep", "keep_offsets", "expected_annot_v_idxs", "expected_annot_ref_pos"):
        d.setdefault(k, None)
    return d


def load_svar2_consensus():
    """tests/golden/pyref_svar2_consensus.npz (make_svar2_fixture.py): decoded two-source channels + the bytes of the reference's
    independent ``_consensus`` (tests/test_svar2_reconstruct.py:66-93) -> list of dicts."""
    z = np.load(GOLDEN / "pyref_svar2_consensus.npz")
    keys = ("ref", "ref_offsets", "regions", "ploidy", "vk_pos", "vk_ilen", "vk_alt_off", "vk_off", "dense_pos", "dense_ilen",
            "dense_alt_off", "dense_range", "dense_present", "dense_present_off", "alt_bytes", "expected", "expected_offsets")
    return [{k: _unbox(z[f"{i}/{k}"]) for k in keys} for i in range(int(z["n"]))]
