"""bench.py's `secondary.annotated_long` / `keep_mask_long` legs on their own.  python tools/long_modes.py"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

print(json.dumps(bench.secondary_long_modes(torch), indent=1))
