#!/usr/bin/env python3
"""GPU box: same-box A/B of library variants built by tools/mkvar.sh.

    python tools/ab_vars.py --vars "noconvr noconvw" --reps 3 [--kbench-reps 20] [--env "WN_PQ_CHAIN=0"]

Each alternation runs `tools/kbench.py bwd` (the fused step's phase times from HIP events, config 2) once per variant, the
shipped library first; prints per-variant medians of every phase and writes gpurun_out/ab_vars_<tag>.json.
"""
import argparse, json, os, statistics, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = "c2"


def run(var, kreps, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    if ":" in var:                                   # "name:ENV=VAL[;ENV=VAL]": the shipped library (or the variant `name`, if built) under these switches
        env.update(dict(kv.split("=", 1) for kv in var.split(":", 1)[1].split(";")))
        lib = os.path.join(ROOT, "tools", "_var_%s.so" % var.split(":", 1)[0])
        if os.path.exists(lib):
            env["WAVENET_HIP_LIB"] = lib
    elif var != "shipped":
        env["WAVENET_HIP_LIB"] = os.path.join(ROOT, "tools", "_var_%s.so" % var)
    if BENCH == "ae":                                # config 4: tools/ae_phases.py prints "X ms/step; phase ms, ..."
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ae_phases.py")], env=env, capture_output=True, text=True, timeout=600)
        for line in reversed(p.stdout.strip().splitlines()):
            if "ms/step;" in line:
                head, rest = line.split(";", 1)
                d = {"step": float(head.split()[0])}
                for kv in rest.split(","):
                    k, v = kv.strip().rsplit(" ", 1)
                    d[k] = float(v)
                return d
        raise RuntimeError("no result for %s: %s" % (var, p.stderr[-400:]))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "bwd", "--reps", str(kreps)], env=env,
                       capture_output=True, text=True, timeout=600)
    for line in reversed(p.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)["phase_ms"]
    raise RuntimeError("no result for %s: %s" % (var, p.stderr[-400:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vars", required=True)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kbench-reps", type=int, default=20)
    ap.add_argument("--env", default="")
    ap.add_argument("--tag", default="ab")
    ap.add_argument("--bench", default="c2", help="c2: tools/kbench.py bwd (config 2 phases); ae: tools/ae_phases.py (config 4 phases)")
    a = ap.parse_args()
    global BENCH
    BENCH = a.bench
    extra = dict(kv.split("=", 1) for kv in a.env.split()) if a.env else {}
    names = ["shipped"] + a.vars.split()
    res = {n: [] for n in names}
    for r in range(a.reps):
        for n in names:
            try:
                res[n].append(run(n, a.kbench_reps, extra))
            except Exception as e:      # a variant that fails must not cost the others their numbers
                print("!!", n, e, flush=True)
    keys = ["stack_fwd", "epilogue_fwd", "softmax_ce", "epilogue_bwd", "stack_bwd", "slab_reduce"]
    if BENCH == "ae":
        keys = ["step", "enc_stack_fwd", "dec_stack_fwd", "ce_epilogue_bwd", "dec_stack_bwd", "enc_stack_bwd"]
    out = {}
    print("%-14s" % "variant" + "".join("%14s" % k for k in keys) + "%10s" % "sum")
    for n in names:
        if not res[n]:
            continue
        med = {k: statistics.median(x[k] for x in res[n]) for k in res[n][0]}
        out[n] = dict(median=med, runs=res[n])
        print("%-14s" % n.split(":")[0] + "".join("%14.3f" % med.get(k, float("nan")) for k in keys) + ("%10.3f" % sum(v for k_, v in med.items() if k_ != "step")), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(env=extra, result=out), open(os.path.join(ROOT, "gpurun_out", "ab_vars_%s.json" % a.tag), "w"), indent=1)


if __name__ == "__main__":
    main()
