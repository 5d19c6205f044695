#!/usr/bin/env python3
"""Development-time transcription of the known answers held by the reference's own tests of
`kepler::propagate_universal` (src/kepler/propagation.rs:217-423 real fink-fat states, :434-880 edge regimes)
into tests/golden/reference_kats_propagation.json — DATA only (inputs, expected outputs, tolerances, the solver
settings of each case), read from the test bodies as text. The expected states are the reference authors'
independent ground truth (scipy DOP853 / mpmath), not outputs of the reference.

Usage (needs /root/reference, i.e. the build container): python tools/transcribe_propagation_kats.py
"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/kepler/propagation.rs"
NUM = r"[-+]?[0-9][0-9_]*\.?[0-9_]*(?:[eE][-+]?[0-9]+)?"


def num(s):
    return float(s.replace("_", ""))


def vec(s):
    return [num(x) for x in re.findall(NUM, s)]


def expr(s, env):
    s = s.strip()
    if s in env:
        return env[s]
    m = re.fullmatch(rf"(\w+)\s*([-+])\s*({NUM})", s)
    if m:
        return env[m.group(1)] + (1 if m.group(2) == "+" else -1) * num(m.group(3))
    return num(s)


def main():
    text = open(SRC).read()
    lines = text.split("\n")
    starts = [i for i, ln in enumerate(lines) if re.match(r"\s*fn test_\w+\(\)", ln)]
    cases, special = [], {}
    for k, i in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(lines)
        name = re.match(r"\s*fn (test_\w+)\(\)", lines[i]).group(1)
        close = next((j for j in range(i, end) if lines[j] == "    }"), end)   # the function's own closing brace
        body = "\n".join(lines[i:close + 1])
        if "proptest" in body or "prop_assert" in body:
            continue
        env = {"T0": 60000.0}
        for m in re.finditer(rf"let (t0|t1)(?:: f64)? = ({NUM});", body):
            env[m.group(1)] = num(m.group(2))
        for m in re.finditer(rf"let (t1) = (T0 [-+] {NUM});", body):
            env[m.group(1)] = expr(m.group(2), env)
        vecs = {m.group(1): vec(m.group(2)) for m in
                re.finditer(r"let (\w+)(?:: Vector3<f64>)?\s*=\s*Vector3::new\(([^;]*?)\);", body, re.S)}
        calls = re.findall(r"propagate_universal\(\s*&position,\s*&velocity,\s*([^,]+?),\s*([^,]+?),\s*([\w()]+)\s*,?\s*\)", body)
        guess = re.search(rf"psi_guess: Some\(({NUM})\)", body)
        src_line = f"src/kepler/propagation.rs:{i + 1}"
        if "expected_r1" in vecs and len(calls) == 1:
            tol = re.search(rf"expected_r1,\s*({NUM})|\(result\.r1 - expected_r1\)\.norm\(\) < ({NUM})", body)
            cases.append({"name": name, "source": src_line, "position": vecs["position"], "velocity": vecs["velocity"],
                          "t0": expr(calls[0][0], env), "t1": expr(calls[0][1], env), "kind": 2,
                          "convergency": 2.220446049250313e-14, "max_iter_prelim_kepuni": 20,
                          "psi_guess": num(guess.group(1)) if guess else None,
                          "expected_r1": vecs["expected_r1"], "expected_v1": vecs["expected_v1"],
                          "tol": num(tol.group(1) or tol.group(2))})
        else:
            inline = [vec(m) for m in re.findall(r"Vector3::new\(([^;]*?)\),\s*\n\s*[0-9e.-]+,", body, re.S)]
            special[name] = {"source": src_line, "position": vecs.get("position"), "velocity": vecs.get("velocity"),
                             "calls": [[expr(a, env) if re.fullmatch(rf"\w+(\s*[-+]\s*{NUM})?", a.strip()) else a.strip(),
                                        expr(b, env) if re.fullmatch(rf"\w+(\s*[-+]\s*{NUM})?|{NUM}", b.strip()) else b.strip()]
                                       for a, b, _ in calls],
                             "inline_expected": inline}
    out = {"source": "src/kepler/propagation.rs tests_propagation (:217-423) and tests_propagation_edge_cases (:434-880); "
                     "SolverKind::Auto, convergency 2.220446049250313e-14, Cardano, max_iter_prelim_kepuni 20 "
                     "(default_solver_type :443-453)",
           "cases": cases, "special": special}
    path = os.path.join(ROOT, "tests", "golden", "reference_kats_propagation.json")
    json.dump(out, open(path, "w"), indent=1)
    print(f"{len(cases)} vector cases, {len(special)} special cases -> {path}")
    for c in cases:
        print(" ", c["name"], c["t1"] - c["t0"], c["tol"], c["psi_guess"])
    for k, v in special.items():
        print("  special", k, v["calls"], len(v["inline_expected"]))


if __name__ == "__main__":
    main()
