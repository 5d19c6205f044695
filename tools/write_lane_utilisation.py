"""profiles/<name>_lane_utilisation.md from profiles/latest_pmc.json (written by tools/summarize_profile.py).
usage: python tools/write_lane_utilisation.py <profile tag, e.g. r04f> <name, e.g. round04f>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
lines = []
for k, r in sorted(pmc["kernels"].items()):
    if k.startswith("at::"):
        continue
    lines.append("%-32s %14.3e %20.3e %22.3e %19.1f" % (k, r["valu_wave_instr"], r["active_inst_valu"], r["thread_cycles_valu"],
                                                       r["thread_cycles_valu"] / max(1.0, r["active_inst_valu"])))
txt = f"""# Lane utilisation of the IOD kernels (build with source hash {pmc['source_hash']})

`tools/profile_gpu.sh {tag} ...` pass `pmc_lanes` (`rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_WAVE_CYCLES -- python3 bench.py --workload headline --steps 1 --warmup 0`; 100 000 objects, one MI355X). Active lanes per VALU
instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (both in quad-cycles), out of 64; per step.

```
kernel                            SQ_INSTS_VALU  SQ_ACTIVE_INST_VALU  SQ_THREAD_CYCLES_VALU  lanes / instruction
""" + "\n".join(lines) + f"""
```
Where the idle lanes are is measured from inside per kernel (stats builds): K2b's solves converge per lane and the wave
waits for its slowest (HISTORY.md section 3, profiles/round05_k2b_isa_account.md); `k_diffcor` and K3b lost theirs in the
range branches of glibc's sincos / atan2 until round 5 selected the operands per lane (37.7 -> 45.7 and 28.9 -> 39.0
lanes; profiles/round05_diffcor_phase_lanes.log). The other workloads' figures (noise points, corrector configurations,
long arcs) are in profiles/{name}_rocprof_summary.md and, per kernel, in the bench detail file's
`roofline.per_kernel[].lanes_per_instruction`. What an instruction costs to issue (4 cycles; TRANS 16) is measured in
profiles/round04_issue_cost_gfx950.json.
"""
open(os.path.join(ROOT, "profiles", f"{name}_lane_utilisation.md"), "w").write(txt)
print(txt)
