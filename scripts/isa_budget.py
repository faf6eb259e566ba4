#!/usr/bin/env python3
"""ISA-counted instruction budget of the pair loop (bench.py's valu_roofline).

    python scripts/isa_budget.py [kernel-name-prefix]

Compiles the device code to assembly (hipcc -S, gfx950), finds the innermost loop of the fast kernel
(k_step_tile<3, LJ, uniform, no energies, no prune> by default) and counts its vector instructions by class:
fp64 full-rate (v_*_f64 except v_rcp), v_rcp_f64 (quarter rate: 4 slots), 32-bit VALU (half a slot: the SIMD
issues a wave64 32-bit op in 2 cycles, an fp64 op in 4), LDS and global memory instructions.  One loop iteration
handles MD_UNROLL = 8 candidates (16 when the compiler unrolls the loop once more: counted from the LDS reads)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "moleculardynamics", "jl_amd", "csrc", "mdhip.hip")
prefix = sys.argv[1] if len(sys.argv) > 1 else "_Z11k_step_tileILi3ELi0ELb1ELb0ELb0E"
out = "/tmp/mdhip_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, SRC],
               check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(tuple(":")) or (l.startswith(prefix) and ":" in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
# The compiler rotates the loop: [P: shared math ... exit branch] [H (header): row/LDS loads, d2, classification ...
# s_cbranch_vccz T] [exact re-decision blocks, rarely run] [T: ... branch back to P].  Hot path = P..H, H..vccz, T..back.
hdrs = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:.*Loop Header", l)]
H = next(h for h in hdrs if any("s_cbranch_vccz" in l for l in body[h:h + 200]))
vccz = next(i for i in range(H, len(body)) if "s_cbranch_vccz" in body[i])
T = body[vccz].split()[-1]
tl = next(i for i, l in enumerate(body) if l.startswith(T + ":"))
P = next(i for i in range(H - 1, 0, -1) if re.match(r"^\.LBB\d+_\d+:", body[i]))
plab = body[P].split(":")[0]
back = next(i for i in range(tl, len(body)) if re.search(r"s_(c)?branch\w*\s+" + re.escape(plab) + r"\b", body[i]))
hot = body[P:H] + body[H:vccz + 1] + body[tl:back + 1]
cls = {"fp64": 0, "rcp64": 0, "valu32": 0, "lds": 0, "vmem": 0, "salu": 0}
for s in hot:
    t = s.strip().split()[0] if s.strip() and not s.strip().startswith((";", ".")) else ""
    if not t:
        continue
    if t.startswith("v_rcp_f64"):
        cls["rcp64"] += 1
    elif re.match(r"v_\w+_f64", t) or t.startswith(("v_lshlrev_b64", "v_lshl_add_u64", "v_mov_b64")):
        cls["fp64"] += 1
    elif t.startswith("v_"):
        cls["valu32"] += 1
    elif t.startswith("ds_"):
        cls["lds"] += 1
    elif t.startswith(("global_", "buffer_", "flat_")):
        cls["vmem"] += 1
    elif t.startswith("s_"):
        cls["salu"] += 1
slots = cls["fp64"] + 4 * cls["rcp64"] + 0.5 * cls["valu32"]
# three ds_read_b64 per candidate (x, y, z): the compiler may have unrolled the source loop (8 candidates) further
cands = cls["lds"] / 3.0
print("kernel", prefix)
print("hot-path loop instructions per machine-loop iteration (%g candidates; the rare exact re-decision blocks excluded):" % cands, cls)
print("fp64-rate issue slots per iteration: %.1f   per 8 candidates: %.1f   per candidate: %.2f" % (slots, 8 * slots / cands, slots / cands))
