"""Known-answer vectors of the SVAR2 two-source provider, transcribed (values only) from the reference's Rust unit tests:

* src/svar2/mod.rs:598-700      decode_alt, merge_hap, hap_diffs_svar2
* src/svar2/mod.rs:702-875      split_to_flat (three tests)
* src/reconstruct/mod.rs:1540-1813   reconstruct_haplotypes_from_svar2 (four tests)
* src/tracks/mod.rs:2480-2567   shift_and_realign_tracks_from_svar2

Keys are symbolic -- ("inline", alt bytes) = encode_alt_inline(alt, 0), ("pure_del", ilen) = encode_pure_del(ilen),
("lookup", row) = encode_lookup(row) -- and are turned into decoded channels by ``oracle.decode_channels`` (decode_alt's three
cases as data): the codec crate itself is not part of the reference's tree.  The same table is replayed through the oracle
(tests/test_oracle_svar2.py) and through the HIP path (tests/test_gpu_svar2.py).
"""

import numpy as np

S = lambda s: np.frombuffer(s, np.uint8).copy()  # noqa: E731

# (name, kwargs of reconstruct_haplotypes_from_svar2_into, initial out bytes, expected out bytes)
RECON_KATS = [
    # src/reconstruct/mod.rs:1573-1632: SNP C->T at 1 in var_key, 1 bp pure DEL at 4 in dense (present)
    ("snp_and_del", dict(ref=b"ACGTACGT", regions=[[0, 0, 8]], shifts=[[0]],
                         vk_pos=[1], vk_keys=[("inline", b"T")], vk_off=[0, 1],
                         dense_pos=[4], dense_keys=[("pure_del", -1)], dense_range=[[0, 1]], dense_present=[0b1],
                         dense_present_off=[0, 1], out_bounds=[[0, 8]]),
     b"NNNNNNNN", b"ATGTAGTN"),
    # :1637-1681: a lone 1 bp pure DEL keeps its anchor base
    ("pure_del_keeps_anchor", dict(ref=b"ACGT", regions=[[0, 0, 4]], shifts=[[0]],
                                   vk_pos=[], vk_keys=[], vk_off=[0, 0],
                                   dense_pos=[1], dense_keys=[("pure_del", -1)], dense_range=[[0, 1]], dense_present=[0b1],
                                   dense_present_off=[0, 1], out_bounds=[[0, 4]]),
     b"NNNN", b"ACTN"),
    # :1687-1740: rows at non-monotonic, gapped destinations; the gap stays untouched
    ("scatter_write", dict(ref=b"ACGT", regions=[[0, 0, 4], [0, 0, 4]], shifts=[[0], [0]],
                           vk_pos=[], vk_keys=[], vk_off=[0, 0, 0],
                           dense_pos=[], dense_keys=[], dense_range=[[0, 0], [0, 0]], dense_present=[],
                           dense_present_off=[0, 0, 0], out_bounds=[[6, 10], [0, 4]]),
     b"----------", b"ACGT--ACGT"),
    # :1757-1812: a zero-length row that shares its start with a non-empty one
    ("scatter_zero_length_tied", dict(ref=b"ACGT", regions=[[0, 0, 4], [0, 2, 2]], shifts=[[0], [0]],
                                      vk_pos=[], vk_keys=[], vk_off=[0, 0, 0],
                                      dense_pos=[], dense_keys=[], dense_range=[[0, 0], [0, 0]], dense_present=[],
                                      dense_present_off=[0, 0, 0], out_bounds=[[4, 8], [4, 4]]),
     b"--------", b"----ACGT"),
]

# src/svar2/mod.rs:598-613
DECODE_KATS = [
    (("pure_del", -2), b"", [0], -2, b""),
    (("lookup", 0), b"ACGT", [0, 4], 3, b"ACGT"),
    (("inline", b"T"), b"", [0], 0, b"T"),          # (the constructor the reconstruct tests use: a 1-base ALT)
]

# src/svar2/mod.rs:615-652: position-sorted, var_key before dense on the tie at 20
MERGE_KAT = dict(vk_pos=[10, 20], vk_key=[100, 200], dense_pos=[15, 20, 30], dense_key=[150, 250, 300], ds=0, de=3,
                 present=[True, True, True], expected=[(10, 100), (15, 150), (20, 200), (20, 250), (30, 300)])

# src/svar2/mod.rs:654-700: SNP at 10 and a 1 bp DEL at 20 inside [0, 100) -> -1
DIFFS_KAT = dict(regions=[[0, 0, 100]], ploidy=1, vk_pos=[10, 20], vk_keys=[("inline", b"A"), ("pure_del", -1)],
                 vk_off=[0, 2], dense_pos=[], dense_keys=[], dense_range=[[0, 0]], dense_present=[], dense_present_off=[0, 0],
                 expected=[[-1]])

# src/tracks/mod.rs:2509-2566: a dense pure DEL of 2 at 1, REPEAT_5P
TRACK_KAT = dict(track=[10.0, 20.0, 30.0, 40.0, 50.0], track_offsets=[0, 5], regions=[[0, 0, 4]], shifts=[[0]],
                 vk_pos=[], vk_keys=[], vk_off=[0, 0], dense_pos=[1], dense_keys=[("pure_del", -2)], dense_range=[[0, 1]],
                 dense_present=[0b1], dense_present_off=[0, 1], out_offsets=[0, 4], params=[0.0], strategy_id=0, base_seed=0,
                 expected=[10.0, 20.0, 50.0, 0.0])

# src/svar2/mod.rs:702-875: split_to_flat.  Each: BatchResultSplit fields -> the flat single-dense-channel layout
SPLIT_KATS = [
    ("marshals_readbound_split", dict(
        n_regions=1, ploidy=1, vk=[(5, 100)], vk_off=[0, 1],
        dense_snp=[(10, 200)], dense_snp_range=[(0, 1)], dense_snp_present=[0b1], dense_snp_present_off=[0, 1],
        dense_indel=[(15, 300)], dense_indel_range=[(0, 1)], dense_indel_present=[0b0], dense_indel_present_off=[0, 1]),
     dict(vk_pos=[5], vk_key=[100], vk_off=[0, 1], dense_pos=[10, 15], dense_key=[200, 300], dense_range=[0, 2],
          dense_present=[0b01], dense_present_off=[0, 2])),
    ("trailing_zero_byte_is_allocated", dict(
        n_regions=12, ploidy=1, vk=[], vk_off=[0] * 13,
        dense_snp=[(42, 7)], dense_snp_range=[(0, 1)] * 12, dense_snp_present=[0b00001001, 0], dense_snp_present_off=list(range(13)),
        dense_indel=[], dense_indel_range=[(0, 0)] * 12, dense_indel_present=[], dense_indel_present_off=[0] * 13),
     dict(vk_pos=[], vk_key=[], vk_off=[0] * 13, dense_pos=[42] * 12, dense_key=[7] * 12,
          dense_range=[v for q in range(12) for v in (q, q + 1)],
          dense_present=[0b00001001, 0b00000000], dense_present_off=list(range(13)))),
    ("ploidy_gt1_reuses_per_query_window", dict(
        n_regions=2, ploidy=2, vk=[], vk_off=[0] * 5,
        dense_snp=[(10, 200), (11, 201), (12, 202), (13, 203)], dense_snp_range=[(0, 2), (2, 4)],
        dense_snp_present=[0b00111001], dense_snp_present_off=[0, 2, 4, 6, 8],
        dense_indel=[(50, 500), (51, 501)], dense_indel_range=[(0, 1), (1, 2)], dense_indel_present=[0b00001101],
        dense_indel_present_off=[0, 1, 2, 3, 4]),
     dict(vk_pos=[], vk_key=[], vk_off=[0] * 5, dense_pos=[10, 11, 50, 12, 13, 51], dense_key=[200, 201, 500, 202, 203, 501],
          dense_range=[0, 3, 3, 6], dense_present=[0b11010101, 0b00001001], dense_present_off=[0, 3, 6, 9, 12])),
]


def kat_channels(oracle, k):
    """The decoded-channel arguments of a KAT dict, in the order the SVAR2 entry points take them (after ``shifts``)."""
    ch = oracle.decode_channels(k["vk_keys"], k["dense_keys"], k.get("lut_bytes", b""), k.get("lut_off", (0,)))
    return dict(vk_pos=np.asarray(k["vk_pos"], np.int32), vk_ilen=ch["vk_ilen"], vk_alt_off=ch["vk_alt_off"],
                vk_off=np.asarray(k["vk_off"], np.int64), dense_pos=np.asarray(k["dense_pos"], np.int32),
                dense_ilen=ch["dense_ilen"], dense_alt_off=ch["dense_alt_off"],
                dense_range=np.asarray(k["dense_range"], np.int32).reshape(-1, 2),
                dense_present=np.asarray(k["dense_present"], np.uint8), dense_present_off=np.asarray(k["dense_present_off"], np.int64),
                alt_bytes=ch["alt_bytes"])
