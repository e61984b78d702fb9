"""tests/spawn_server.py: the helper that forks the tests' child processes so that the test process (HIP runtime loaded) never does."""
import subprocess
import sys

import pytest
import spawn_server


def test_the_helper_is_running_and_is_not_this_process():
    assert spawn_server.active()
    out = spawn_server.run([sys.executable, "-c", "import os; print(os.getppid())"], capture_output=True, text=True, check=True).stdout
    assert int(out) == spawn_server._proc.pid  # the child's parent is the helper, not pytest


def test_results_and_exceptions_come_back_as_subprocess_run_would_give_them(tmp_path):
    res = spawn_server.run([sys.executable, "-c", "import sys; print('out'); print('err', file=sys.stderr); sys.exit(3)"], capture_output=True, text=True)
    assert (res.returncode, res.stdout.strip(), res.stderr.strip()) == (3, "out", "err")
    with pytest.raises(subprocess.CalledProcessError) as bad:
        spawn_server.run([sys.executable, "-c", "import sys; sys.exit(4)"], check=True, capture_output=True)
    assert bad.value.returncode == 4
    with pytest.raises(subprocess.TimeoutExpired):
        spawn_server.run([sys.executable, "-c", "import time; time.sleep(30)"], timeout=0.5)
    # environment and working directory travel; an UNCAPTURED child must not write into the reply pipe
    res = spawn_server.run([sys.executable, "-c", "print('noise on stdout')"])
    assert res.returncode == 0
    res = spawn_server.run([sys.executable, "-c", "import os; print(os.environ['GYMRS_SPAWN_TEST'], os.getcwd())"], env={"GYMRS_SPAWN_TEST": "x"},
                           cwd=tmp_path, capture_output=True, text=True, check=True)
    assert res.stdout.split() == ["x", str(tmp_path)]
    with pytest.raises(FileNotFoundError):
        spawn_server.run(["/nonexistent/binary"])
