"""Host-side logic of the numpy drop-in layer (no GPU): the per-dataset cache key."""
import gc

import numpy as np


def test_cache_key_fingerprints_read_only_arrays_once(monkeypatch):
    import genvarloader_amd.ffi as ffi

    calls = []
    real = ffi._fingerprint
    monkeypatch.setattr(ffi, "_fingerprint", lambda a: (calls.append(a.shape), real(a))[1])
    big = np.arange(1 << 19, dtype=np.int32)          # 2 MiB: the sampled fingerprint
    big.flags.writeable = False                        # what np.memmap(..., mode="r") hands over
    k1, k2 = ffi._key(big), ffi._key(big)
    assert k1 == k2 and len(calls) == 1                # remembered per object: the second call touches nothing
    n_cached = len(ffi._FP_CACHE)
    view = big[:]                                      # another object over the same memory: its own entry, same key
    assert ffi._key(view) == k1 and len(calls) == 2
    del view
    gc.collect()
    assert len(ffi._FP_CACHE) == n_cached              # entries die with their arrays
    w = np.arange(1 << 19, dtype=np.int32)
    ka = ffi._key(w)
    w[::1024] += 1                                     # a writable array is looked at on every call
    assert ffi._key(w) != ka and len(calls) == 4
    small = np.arange(100, dtype=np.int64)
    ks = ffi._key(small)
    small[57] = -1                                     # below 1 MiB the fingerprint is exact
    assert ffi._key(small) != ks
