"""Short randomised sweeps (tools/fuzz.py, tools/fuzz_tracks.py): random contigs, variant
densities, indel sizes, ploidy 1-3, lengths 1-5000, ragged / fixed, shifts far beyond
max_shift, keep masks, regions stride 3 / 4, annotations, both one-hot layouts, all five
insertion-fill strategies, overlapping intervals -- HIP vs oracle, bit-exact.  Once on the
planned paths (scan-free / packed / per-wave scans, and each of them switched off in turn) and
once with every row forced through the scalar path."""

import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _run(script, n, seed, dbg=None, **extra_env):
    env = dict(os.environ)
    if dbg is not None:
        env["GVL_DBG"] = str(dbg)
    env.update({k: str(v) for k, v in extra_env.items()})
    r = subprocess.run([sys.executable, str(REPO / "tools" / script), str(n), str(seed)], capture_output=True,
                       text=True, env=env, cwd=REPO, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_fuzz_haplotypes_planned_path():
    _run("fuzz.py", 600, 101)


def test_fuzz_haplotypes_scalar_path():
    _run("fuzz.py", 300, 102, dbg=8)


def test_fuzz_haplotypes_without_scan_free_plan():
    """GVL_DBG=32: SNP-only rows go through the packed plan like rows with indels."""
    _run("fuzz.py", 300, 106, dbg=32)


def test_fuzz_haplotypes_per_wave_scans_only():
    """GVL_DBG=512: no packable rows, every row runs the per-wave scans."""
    _run("fuzz.py", 300, 107, dbg=512)


def test_fuzz_haplotypes_without_slot_records():
    """GVL_DBG=64: ignore gvl_static.slot_rec: rows find their records through the CSR + geno_rec."""
    _run("fuzz.py", 300, 108, dbg=64)


def test_fuzz_haplotypes_without_genotype_records():
    """GVL_DBG=80: ignore slot_rec and geno_rec, i.e. the geno_v_idxs -> vrec gather."""
    _run("fuzz.py", 300, 105, dbg=80)


def test_fuzz_haplotypes_without_speculative_reads():
    """GVL_DBG=128: reference bytes are requested after the plan only."""
    _run("fuzz.py", 200, 109, dbg=128)


def test_fuzz_tracks_planned_walk():
    _run("fuzz_tracks.py", 400, 103)


def test_fuzz_tracks_scalar_walk():
    _run("fuzz_tracks.py", 200, 104, dbg=8)


def test_fuzz_lean_kernel():
    """One-hot only, fixed-length rows of at most 2048 bases: recon_lean_kernel (nibble-packed reference, plan on
    the row's record lanes) and, for the rows it hands over, its solo general path."""
    _run("fuzz_lean.py", 600, 201)


def test_fuzz_lean_kernel_every_row_solo():
    """GVL_DBG=32768: the lean kernel hands EVERY row to the all-purpose body in SOLO mode."""
    _run("fuzz_lean.py", 300, 202, dbg=32768)


def test_fuzz_lean_kernel_rereads_indel_rows():
    """GVL_DBG=65536: rows with indels re-read their runs from the packed reference instead of re-aligning the window in LDS."""
    _run("fuzz_lean.py", 300, 203, dbg=65536)


@pytest.mark.parametrize("dbg", [33554432, 33554432 + 65536, 33554432 + 32768],
                         ids=["pipelined", "pipelined-defers-indel-rows", "pipelined-defers-every-row"])
def test_fuzz_lean_kernel_pipelined(dbg):
    """GVL_DBG & 33554432: the lean kernel's pipelined form (gvl_lean_pipe.inc) on ONE workgroup -- a wave takes every fourth
    row, up to 30 rows per wave: window + slot line by LDS-DMA a row ahead, rows it cannot take run at the wave's end."""
    _run("fuzz_lean.py", 400, 204 + (dbg >> 15) % 7, dbg=dbg)


@pytest.mark.parametrize("dbg,sub", [(0, 2), (32768, 2), (65536, 2), (0, 1), (0, 4), (536870912, 2), (536870912, 1)],
                         ids=["default", "every-chunk-solo", "rereads", "one-chunk-per-wave", "four-chunks-per-wave", "no-chunk-plans", "no-chunk-plans-one-chunk-per-wave"])
def test_fuzz_lean_kernel_long_rows(dbg, sub):
    """FUZZ_LONG=1: rows of 2 052 ... 40 000 bases = the lean kernel's chunked form (BASELINE config 4's haplotype kernel),
    with 1 / 2 / 4 consecutive chunks per wave (FUZZ_SUB -> gvl_set_tuning)."""
    _run("fuzz_lean.py", 120, 210 + sub + (dbg >> 15), dbg=dbg, FUZZ_LONG=1, FUZZ_SUB=sub)


@pytest.mark.parametrize("dbg", [0, 2097152, 1073741824], ids=["default", "no-window", "general-track-kernel"])
def test_fuzz_tracks_straight_from_intervals(dbg):
    """tools/fuzz_fused_tracks.py: realign_tracks_kernel<PAINT> (BASELINE config 4's track kernel): a batch's tracks realigned
    straight from their intervals, `tile_complete` interval sets."""
    _run("fuzz_fused_tracks.py", 150, 220, dbg=dbg)


@pytest.mark.parametrize("dbg,sub", [(0, 2), (32768, 2), (65536, 2), (0, 1), (1048576, 2)],
                         ids=["default", "every-chunk-solo", "rereads", "one-chunk-per-wave", "all-purpose-kernel"])
def test_fuzz_lean_kernel_ragged_long_rows(dbg, sub):
    """FUZZ_LONG=1 FUZZ_RAGGED=1: ragged rows (output_length = -1) of 2 052 ... 40 000 bases = recon_lean_kernel<.., LONG, RAGL>."""
    _run("fuzz_lean.py", 120, 230 + sub + (dbg >> 15), dbg=dbg, FUZZ_LONG=1, FUZZ_RAGGED=1, FUZZ_SUB=sub)


def test_fuzz_lean_kernel_ragged_short_rows():
    """FUZZ_RAGGED=1: ragged rows of at most 2 048 + 6 bases = recon_lean_rows_kernel's ragged form."""
    _run("fuzz_lean.py", 300, 240, FUZZ_RAGGED=1)


@pytest.mark.parametrize("ragged", [0, 1], ids=["fixed", "ragged"])
def test_fuzz_lean_kernel_many_batches_one_grid(ragged):
    """FUZZ_MANY=1: 4-16 batches of 1 500-6 000 queries in ONE multi-workgroup grid on default flags (what bench.py times and the
    native loader launches), 1 / 1.5 / 2 / 3 / 8 rows per wave, every batch of every launch against the oracle."""
    _run("fuzz_lean.py", 24, 250 + ragged, FUZZ_MANY=1, FUZZ_RAGGED=ragged)


@pytest.mark.parametrize("dbg", [0, 256], ids=["front-workgroups", "solo-at-the-waves-ends"])
def test_fuzz_lean_kernel_mixed_ragged_batches(dbg):
    """FUZZ_MIXED=1: ragged batches of mostly short rows with a few of 2 600 ... 20 000 bases on the pipelined kernel (round 6): the long
    rows by the launch's front workgroups, chunks in parallel / by the wave that meets them."""
    _run("fuzz_lean.py", 200, 260 + (dbg >> 8), dbg=dbg, FUZZ_MIXED=1)


def test_fuzz_svar2_provider():
    """tools/fuzz_svar2.py: random two-source batches (windows across the packed form's 16 entries and the general form's 64-entry
    tiles, filter_exonic, ragged / fixed, shifts, RC) -- gvl_svar2_merge + the kernels over the merged table vs the oracle's provider."""
    _run("fuzz_svar2.py", 250, 201)


def test_fuzz_svar2_provider_csr_routes():
    """... with the slot lines ignored (GVL_DBG 64: records through the merged table's CSR) and through the vrec gather (80)."""
    _run("fuzz_svar2.py", 120, 202, dbg=64)
    _run("fuzz_svar2.py", 120, 203, dbg=80)
