"""A short session of tools/fuzz_parity.py under pytest: random scene files (tests/synth.py's generators with random
counts, seeds, cameras and frame sizes) through the C ABI against the oracle — f64 first hits bit for bit, f32 first
hits within their budget, six frames per scene; over scene files, over the example generators, and over scenes around a large random mesh.  The long
sessions (3 x 3000 scenes) are in
profiles/r3_fuzz_parity.txt.  Needs an MI355X:  pytest -m gpu."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,n", [([], 96), (["--examples"], 96), (["--meshes", "--extras"], 9)],
                         ids=["scene files", "example generators", "large meshes with extras"])
def test_random_scenes_against_the_oracle(mode, n):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--scenes", str(n), "--seed", "11"] + mode,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert f"{n} scenes" in out.stdout and "0 with a missed bar" in out.stdout
