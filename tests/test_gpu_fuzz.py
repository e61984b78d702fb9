"""A short run of tools/fuzz_engine_vs_twin.py inside the suite: random operation sequences over all env kinds and flag sets --
gymrs_step_many calls of random length (HIP launches and chains), single steps, fused rollouts, resets, set_state, set_params with
a new episode cap, clones -- each operation replayed by the CPU f32 twin and compared bit for bit.  (The long runs are in
profiles/r03_fuzz_engine_vs_twin.log.)"""
import spawn_server
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("seed", [11, 12])
def test_random_operation_sequences_match_the_twin(seed):
    res = spawn_server.run([sys.executable, str(ROOT / "tools" / "fuzz_engine_vs_twin.py"), "--cases", "20", "--seed", str(seed), "--ops", "12",
                          "--max-lanes", "40000"], cwd=str(ROOT), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "fuzz ok:" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
