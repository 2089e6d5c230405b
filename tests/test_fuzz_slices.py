"""Seeded slices of the hand-run fuzzers, collected so that every `pytest -m gpu` run sees them:
  tests/fuzz_oracle.py            operators through the C-ABI (default routing) vs the fp64 oracle
  tools/fuzz_fastpaths.py         default routing vs the generic kernels (dims, orders, bounds, dtypes, zooms, rough fields)
  tools/fuzz_scatter_variants.py  interpol_push_bricks and the owner-computes scatter vs the generic kernels
  tests/sweep_many_tiles*.py      (round 6: were tools/r5/sweep_big*.py) default routing vs the generic kernels where a workgroup serves
                                  several tiles / bricks: every operator, smooth and rough fields, mixed orders, displacement and separable
                                  grids, shared targets, float64 / float16 storage, 300 small batch items -- trimmed by SWEEP_TRIM=1
Each runs as its own process (the scripts are also command-line tools: `python <script> [n_cases] [seed]`)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("script,n_cases,seed", [
    ("tests/fuzz_oracle.py", 60, 2025),
    ("tools/fuzz_fastpaths.py", 120, 2025),
    ("tools/fuzz_scatter_variants.py", 100, 2025),
])
def test_fuzz_slice(script, n_cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), str(n_cases), str(seed)], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("script,env", [
    ("tests/sweep_many_tiles.py", {"SWEEP_TRIM": "1"}),
    ("tests/sweep_many_tiles.py", {"SWEEP_TRIM": "1", "SWEEP_MANY": "1"}),
    ("tests/sweep_many_tiles_forms.py", {"SWEEP_TRIM": "1"}),
    ("tests/sweep_many_tiles_dtypes.py", {}),
])
def test_many_tiles_sweep(script, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, script)], cwd=ROOT, env={**os.environ, **env},
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=850)
    assert r.returncode == 0 and "BAD" not in r.stdout, r.stdout[-3000:]
