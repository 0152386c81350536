"""Per-launch HBM-side traffic of the flash kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).
usage: pmc_traffic.py <fetch_csv> <write_csv> <kernel substring> [<uncounted substring>] -> JSON.  Dispatches matching the
uncounted substring add their bytes but are not launches of their own (the gated exact flash kernel behind the speculative one).  Units/corrections per MI355X_MICROARCH.md (HBM section):
the counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads are doubled; WRITE_SIZE is taken
as is (uncalibrated).  Infinity-Cache hits are included in FETCH_SIZE (fabric-side counter)."""
import csv
import json
import sys


UNCOUNTED = sys.argv[4] if len(sys.argv) > 4 else None


def total(path, counter, sub):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 0 if UNCOUNTED and UNCOUNTED in r["Kernel_Name"] else 1
    return tot, n


f, nf = total(sys.argv[1], "FETCH_SIZE", sys.argv[3])
w, nw = total(sys.argv[2], "WRITE_SIZE", sys.argv[3])
print(json.dumps({"kernel": sys.argv[3], "launches": nf, "fetch_bytes_per_launch_raw": f * 1024 / max(nf, 1),
                  "fetch_bytes_per_launch_corrected_x2": 2 * f * 1024 / max(nf, 1), "write_bytes_per_launch": w * 1024 / max(nw, 1),
                  "traffic_bytes_per_launch": (2 * f + w) * 1024 / max(nf, 1)}))
