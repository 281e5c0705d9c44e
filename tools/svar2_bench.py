"""What the SVAR2 provider costs on the device (bench.py's `secondary.svar2` leg on its own): gvl_svar2_merge over a group of 16
cfg3-shaped batches, gvl_reconstruct_many over the merged table, the same haplotypes through the SVAR1 table, and the pipelined
schedule.  python tools/svar2_bench.py [batches per group] [queries per batch]"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
BQ = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
print(json.dumps(bench.secondary_svar2(torch, G, BQ), indent=1))
