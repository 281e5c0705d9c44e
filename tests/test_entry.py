"""The driver's build check: __graft_entry__.build() must succeed on a CPU-only box (hipcc
cross-compiles gfx950; the library loads and reports the ABI version the binding expects)."""


def test_graft_entry_build():
    import __graft_entry__ as g

    g.build()
