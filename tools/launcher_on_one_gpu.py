#!/usr/bin/env python3
"""The multi-rank launcher path of bench.py, exercised with the ONE GPU a build box has.

This process never imports ibo_amd and makes no HIP call: it only starts `python3 bench.py ...` children, as the
driver would.  With IBO_BENCH_FORCE_SPAWN=1 bench.py takes its self-launch branch even for --gpus 1: a fresh rank
process, a private 0700 rendezvous directory with the RCCL id file, ncclCommInitRank, RCCL barriers and the RCCL
arg-max exchange -- everything the 8-GPU run does except a second rank.  Checks:
  * launcher == "self (bench.py children)", rccl_nranks == 1
  * same arg-max (index and value) as the plain single-process run, `value` within 2 % of it
  * the private rendezvous directory is gone afterwards
  * the sharded gallery / sharded NLML grid run under the launcher too (--gpus 1 without --no-extras)
Exit status non-zero on any failure.   python3 tools/launcher_on_one_gpu.py [out.json]"""
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(extra_env, args):
    env = dict(os.environ, **extra_env)
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    if p.returncode != 0:
        sys.stderr.write(p.stderr.decode("utf-8", "replace")[-4000:])
        raise SystemExit("bench.py %s exited with %d" % (" ".join(args), p.returncode))
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    if len(lines) != 1:
        raise SystemExit("expected ONE JSON line, got %d" % len(lines))
    return json.loads(lines[0])


def main():
    tmpdir = tempfile.gettempdir()
    before = set(glob.glob(os.path.join(tmpdir, "ibo_bench_*")))
    quick = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    plain = run({}, quick)
    spawn = run({"IBO_BENCH_FORCE_SPAWN": "1"}, quick)
    report = {"plain": {k: plain[k] for k in ("value", "ms_per_step", "launcher", "rccl_nranks", "best")},
              "spawned": {k: spawn[k] for k in ("value", "ms_per_step", "launcher", "rccl_nranks", "best")}}
    assert plain["launcher"] == "single process", plain["launcher"]
    assert spawn["launcher"] == "self (bench.py children)", spawn["launcher"]
    assert spawn["rccl_nranks"] == 1, spawn["rccl_nranks"]
    assert spawn["best"] == plain["best"], (spawn["best"], plain["best"])
    assert spawn["best"]["index"] == 29258, spawn["best"]
    ratio = spawn["value"] / plain["value"]
    report["value_ratio_spawned_over_plain"] = ratio
    assert abs(ratio - 1.0) < 0.02, ratio
    # the sharded gallery and NLML grid through the same path (world of one rank, RCCL collectives and all)
    full = run({"IBO_BENCH_FORCE_SPAWN": "1"}, ["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert full["launcher"] == "self (bench.py children)" and full["rccl_nranks"] == 1
    cfg = full["configs"]
    report["spawned_configs"] = {"c3_gallery8_ms": cfg["c3_gallery8"]["ms"], "c3_gallery_min_dist": cfg["c3_gallery8"]["min_pairwise_distance"],
                                 "c5_ms_per_theta": cfg["c5_nlml_grid"]["ms_per_theta"], "c5_argmin": cfg["c5_nlml_grid"]["argmin"],
                                 "c3_shard_sweep_evals_per_s": cfg["c3_shard_sweep"]["value"]}
    assert cfg["c3_gallery8"]["min_pairwise_distance"] > 0.5 and cfg["c5_nlml_grid"]["n_not_pd"] == 0
    for cname in ("c3", "c5"):
        r = run({"IBO_BENCH_FORCE_SPAWN": "1"}, ["--gpus", "1", "--steps", "2", "--warmup", "1", "--config", cname, "--no-cpu-baseline"])
        assert r["launcher"] == "self (bench.py children)" and r["rccl_nranks"] == 1
        report["spawned_config_" + cname] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"]}
    after = set(glob.glob(os.path.join(tmpdir, "ibo_bench_*")))
    report["rendezvous_dirs_left_behind"] = sorted(after - before)
    assert not (after - before), after - before
    report["ok"] = True
    s = json.dumps(report, indent=1)
    print(s)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(s + "\n")


if __name__ == "__main__":
    main()
