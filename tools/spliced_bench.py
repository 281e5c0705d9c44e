"""bench.py's `secondary.spliced` leg on its own.  python tools/spliced_bench.py [pairs per batch] [longest exon] [exonic keep mask 0/1] [GVL_TUNE_MIXED_MIN_ROWS] [GVL_TUNE_PIPE_ROWS_X100]"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

from genvarloader_amd import _lib  # noqa: E402

if len(sys.argv) > 4:
    _lib.set_tuning(_lib.TUNE_MIXED_MIN_ROWS, int(sys.argv[4]))
if len(sys.argv) > 5:
    _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, int(sys.argv[5]))
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
max_exon = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
exonic = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
d = bench.secondary_spliced(torch, pairs=pairs, max_exon=max_exon, exonic=exonic)
d.pop('how')
print(json.dumps(d, indent=1))
