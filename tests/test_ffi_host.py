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


def test_cache_key_real_memmap_and_read_only_views_over_writable_arrays(monkeypatch, tmp_path):
    """What the reference actually passes is np.memmap(mode="r") (_haps.py:435-460): np.asarray of a memmap is a NEW
    base-class array on every call, so the once-per-object cache must key on the memmap itself.  And a read-only VIEW over a
    writable array is not immutable (the base can change under it): fingerprinted on every call."""
    import genvarloader_amd.ffi as ffi

    calls = []
    real = ffi._fingerprint
    monkeypatch.setattr(ffi, "_fingerprint", lambda a: (calls.append(a.shape), real(a))[1])
    f = tmp_path / "v.npy"
    np.arange(1 << 19, dtype=np.int32).tofile(f)
    mm = np.memmap(f, dtype=np.int32, mode="r")
    assert not mm.flags.writeable
    k1, k2, k3 = ffi._key(mm), ffi._key(mm), ffi._key(mm)
    assert k1 == k2 == k3 and len(calls) == 1          # the memmap is looked at once, not on every call
    base = np.arange(1 << 19, dtype=np.int32)
    ro = base[:]
    ro.flags.writeable = False
    ka = ffi._key(ro)
    base[::1024] += 1                                   # an edit through the writable base ...
    assert ffi._key(ro) != ka and len(calls) == 3       # ... is seen: such a view is fingerprinted every time
    del mm
    gc.collect()
