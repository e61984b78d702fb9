"""Children of the test process are started by a helper that never loads the HIP runtime.

`fork()` in a process whose HIP / HSA runtime is up (helper threads, signal handlers, queues mapped from the driver) is the one thing the
round-4 GPU suite ever died of: one run in ~50 ended with a segmentation fault inside `subprocess.run` ("Suite soak"), never
reproduced under a native-backtrace handler.  `start()` is called when conftest.py is imported -- before torch is, so before any HIP
call -- and forks ONE plain python child; `run()` has `subprocess.run`'s signature and hands the command to that child, which runs it
with `subprocess.run` and sends the result (or the exception) back.  Without `start()` (or if the helper died) `run()` is
`subprocess.run`.
"""
import atexit
import pickle
import struct
import subprocess
import sys
import threading

_SERVER = r'''
import pickle, struct, subprocess, sys
inp, out = sys.stdin.buffer, sys.stdout.buffer
while True:
    head = inp.read(8)
    if len(head) < 8:
        break
    args, kw = pickle.loads(inp.read(struct.unpack("<Q", head)[0]))
    kw.setdefault("stdin", subprocess.DEVNULL)
    if not kw.get("capture_output"):
        kw.setdefault("stdout", 2)  # this process's stdout is the reply pipe: an uncaptured child writes to stderr
    try:
        r = subprocess.run(*args, **kw)
        res = ("ok", r.args, r.returncode, r.stdout, r.stderr)
    except subprocess.TimeoutExpired as e:
        res = ("timeout", e.cmd, e.timeout, e.output, e.stderr)
    except subprocess.CalledProcessError as e:
        res = ("called", e.cmd, e.returncode, e.output, e.stderr)
    except BaseException as e:
        try:
            res = ("raised", pickle.dumps(e), None, None, None)
        except Exception:
            res = ("error", repr(e), None, None, None)
    blob = pickle.dumps(res)
    out.write(struct.pack("<Q", len(blob)))
    out.write(blob)
    out.flush()
'''

_proc = None
_lock = threading.Lock()


def start():
    global _proc
    if _proc is None:
        _proc = subprocess.Popen([sys.executable, "-c", _SERVER], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
        atexit.register(stop)
    return _proc


def stop():
    global _proc
    p, _proc = _proc, None
    if p is not None:
        try:
            p.stdin.close()
            p.wait(timeout=5)
        except Exception:
            p.kill()


def active():
    return _proc is not None and _proc.poll() is None


def run(*args, **kw):
    if not active():
        return subprocess.run(*args, **kw)
    blob = pickle.dumps((args, kw))
    with _lock:
        _proc.stdin.write(struct.pack("<Q", len(blob)))
        _proc.stdin.write(blob)
        _proc.stdin.flush()
        head = _proc.stdout.read(8)
        if len(head) < 8:
            raise RuntimeError("the spawn helper went away while running %r" % (args[0] if args else kw.get("args"),))
        kind, a, b, out, err = pickle.loads(_proc.stdout.read(struct.unpack("<Q", head)[0]))
    if kind == "ok":
        return subprocess.CompletedProcess(a, b, out, err)
    if kind == "timeout":
        raise subprocess.TimeoutExpired(a, b, output=out, stderr=err)
    if kind == "called":
        raise subprocess.CalledProcessError(b, a, output=out, stderr=err)
    if kind == "raised":
        raise pickle.loads(a)  # (FileNotFoundError and the like: what subprocess.run itself would have raised)
    raise RuntimeError("the spawn helper could not run the command: " + a)
