"""world_size-2 gloo run of the node-partitioned trainer on CPU (kernels replaced
by tests/cpu_backend.py): the partitioned epochs must reproduce the
single-process epochs -- identical dropout masks through global-id RNG keys,
all-gathered tables, one all-reduce of the flat gradient buffer."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, epochs, drop, out, port):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r),
                               str(world), "cpu", str(epochs), str(drop), out, str(port), "1"],
                              env=env, cwd=ROOT) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0


@pytest.mark.parametrize("drop,port", [(0.0, 29701), (0.6, 29711)])
def test_two_gloo_ranks_match_single_process(tmp_path, drop, port):
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    _launch(1, 2, drop, one, port)
    _launch(2, 2, drop, two, port + 2)
    a, b = np.load(one), np.load(two)
    assert np.isfinite(a["flat"]).all()
    assert np.abs(a["flat"] - b["flat"]).max() < 1e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 1e-5
