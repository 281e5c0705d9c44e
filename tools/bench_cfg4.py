"""Moved to the repo root (``bench_cfg4.py``, next to ``bench.py`` whose default run uses it); kept so that the tools here import as before."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_cfg4 import *  # noqa: F401,F403
from bench_cfg4 import build, measure, main  # noqa: F401
