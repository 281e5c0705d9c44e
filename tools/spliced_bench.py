"""bench.py's `secondary.spliced` leg on its own.  python tools/spliced_bench.py"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = bench.secondary_spliced(torch, pairs=pairs)
d.pop('how')
print(json.dumps(d, indent=1))
