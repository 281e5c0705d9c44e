"""The stand-alone track entry points of BASELINE config 4 (bench.py's `secondary.cfg4.kernels`), with gvl_tracks_batch's sizing inside
the row-plan launch (default) and in a launch of its own (GVL_DBG 4096: round 5's three dependent launches).  python tools/tracks_batch_ab.py"""
import json
import sys
from pathlib import Path
from types import SimpleNamespace

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench_cfg4  # noqa: E402
from genvarloader_amd import _lib  # noqa: E402

A = SimpleNamespace(gpus=1, steps=20, warmup=5, min_region_ms=200.0, max_regions=40, samples=0)
lib = _lib.load()
for flags in (-1, 4096, -1, 4096):
    lib.gvl_set_debug_flags(flags)
    k = bench_cfg4.measure(A, init_dist=False)["kernels"]
    print(flags, json.dumps({n: round(v["ms"] * 1e3, 2) for n, v in k.items() if isinstance(v, dict)}), flush=True)
lib.gvl_set_debug_flags(-1)
