"""
A process that never touches the GPU and starts rank processes on behalf of the GPU tests.

Why it exists: on the GPU pool a process that has initialised the GPU must not exec another program, and a pytest
process that has run GPU tests is such a process -- a fork of it inherits the driver's state.  tests/conftest.py starts
THIS helper in pytest_configure, before any test runs; it imports nothing but the standard library, makes no HIP call,
and stays idle on its stdin.  A test that needs fresh rank processes (tests/test_gpu_two_ranks.py) sends one JSON line

    {"argv": [...], "world": W, "env": {...}, "timeout": seconds, "rank_env": true}

and gets one back: {"rc": [...], "out": [...], "err": [...]} (per rank).  Every child is started with RANK / LOCAL_RANK /
WORLD_SIZE set (unless rank_env is false: a program that starts its own ranks, bench.py --gpus N); if one rank exits non-zero or the time is up, the others are terminated by their exact pids (never by
pattern) and their exit codes reported as they are.  EOF on stdin ends the helper.
"""
import json
import os
import subprocess
import sys
import tempfile
import time


def run(job):
    world = int(job.get("world", 1))
    timeout = float(job.get("timeout", 600))
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ)
        env.update({k: str(v) for k, v in job.get("env", {}).items()})
        if job.get("rank_env", True):
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world))
        fo, fe = tempfile.TemporaryFile(), tempfile.TemporaryFile()
        files.append((fo, fe))
        procs.append(subprocess.Popen([sys.executable] + list(job["argv"]), env=env, stdout=fo, stderr=fe, stdin=subprocess.DEVNULL))
    t0 = time.time()
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        if any(rc not in (None, 0) for rc in rcs) or time.time() - t0 > timeout:
            time.sleep(1.0)                              # let the others notice a closed socket by themselves first
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            break
        time.sleep(0.05)
    res = {"rc": [p.returncode for p in procs], "out": [], "err": [], "seconds": time.time() - t0}
    for fo, fe in files:
        for f, key in ((fo, "out"), (fe, "err")):
            f.seek(0)
            res[key].append(f.read().decode("utf-8", "replace")[-200000:])      # (a bench line is ~10 KB)
            f.close()
    return res


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            reply = run(json.loads(line))
        except Exception as e:                           # the test sees the reason instead of a hung pipe
            reply = {"rc": [], "out": [], "err": [repr(e)], "seconds": 0.0}
        sys.stdout.write(json.dumps(reply) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
