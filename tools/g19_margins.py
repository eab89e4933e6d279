"""Measured margins of fixture G19 (heavy-tailed BatchNorm statistics, tests/test_gpu_heavy_tailed.py) in every matrix mode.
    python tools/g19_margins.py > profiles/r04_g19_margins.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_heavy_tailed import run_g19  # noqa: E402

for mode in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['f16x3', 'bf16x6', 'f32']):
    print(mode, json.dumps(run_g19(mode), indent=1), flush=True)
