#!/usr/bin/env python3
"""pmc_summary.txt (FETCH_SIZE / WRITE_SIZE passes, rocprofv3 units: KiB) -> the traffic JSON bench.py reads.

Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports exactly half the bytes of wide coalesced
streaming reads (16 B per lane) -- doubled here, which over-corrects the narrower row / index reads (an upper
bound on the read side); WRITE_SIZE is exact for wide stores.  The kernel-source hash ties the number to the code
it was measured on."""
import json, re, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
txt = open(sys.argv[1]).read()
def grab(counter, kernel="k_step_tile<3, 0, true, false, false>"):
    m = re.search(re.escape(kernel) + r"\n(?:   .*\n)*?   " + counter + r"\s+n=\s*(\d+)\s+mean=([0-9.e+]+)", txt)
    return (int(m.group(1)), float(m.group(2))) if m else (0, None)
nf, fetch = grab("FETCH_SIZE")
nw, write = grab("WRITE_SIZE")
out = {
    "kernel": "k_step_tile<3, LJ, uniform, no energies, no prune> (ordinary fused step), N = 1048576",
    "launches_sampled": nf,
    "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
    "read_bytes_raw": fetch * 1024 if fetch else None, "write_bytes": write * 1024 if write else None,
    "read_bytes_corrected_x2": 2 * fetch * 1024 if fetch else None,
    "traffic_bytes_raw": (fetch + write) * 1024 if fetch and write else None,
    "traffic_bytes_corrected": (2 * fetch + write) * 1024 if fetch and write else None,
    "algorithmic_bytes": (380.0 - 32.0) * 1048576,   # the kernel's share: SURVEY.md 8(d)'s NVT figure minus the cell binning
    "algorithmic_bytes_whole_step": 380.0 * 1048576,
    "note": "the x2 read correction is the guide's for 16-B-per-lane streaming reads (the record gathers); the rows are read "
            "8 B per lane, where it is uncalibrated: corrected = upper bound, raw = lower bound of the bytes that left L2",
    "kernel_sources_sha256_16": bench.kernel_hash(),
    "recipe": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --steps 20 --warmup 5 --equil 60` (scripts/profile_round.sh)",
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
