#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/prof_round.sh: HBM bytes per launch of the dominant kernel
(conv_wino2_kernel, all template instances of a step pooled), per config.

    python tools/traffic_from_pmc.py gpurun_out/pmc_r02a_ r02a > profiles/traffic.json

Corrections exactly as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes for gfx950: FETCH_SIZE / WRITE_SIZE are in
KiB; FETCH_SIZE reports half of the bytes of a wide (16 B/lane) coalesced read stream -- which is what this kernel's
16-byte LDS-DMA and weight loads are -- so it is doubled; WRITE_SIZE is used as reported (uncalibrated)."""
import csv, glob, json, re, sys
from collections import defaultdict


def per_launch(prefix, ps, counter):
    files = glob.glob(f"{prefix}{ps}/**/*counter_collection.csv", recursive=True)
    tot, disp, dur = defaultdict(float), defaultdict(set), defaultdict(float)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if r["Counter_Name"] != counter:
                continue
            tot[k] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp[k]:
                disp[k].add(r["Dispatch_Id"])
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    ks = [k for k in tot if "conv_wino" in k or "conv_wh" in k or "conv_h2" in k]
    n = sum(len(disp[k]) for k in ks)
    return (sum(tot[k] for k in ks) / n if n else 0.0), n, (sum(dur[k] for k in ks) / n if n else 0.0)


def main(prefix, tag):
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sinddm_amd import build as _build
    lib_hash = _build.stamp()          # what the passes ran: bench.py flags the record stale when the library changes
    alg = {"c2": 16 * 186 * 248, "c3": 64 * 411 * 512}          # pixels per launch
    # algorithmic bytes of the seven Winograd launches of a step: input once + output once, fp32
    chans = [(80, 80), (80, 160), (160, 160), (160, 160), (160, 160), (160, 80), (80, 80)]
    out = {}
    for cfg in ("c2", "c3"):
        fetch, n, dur = per_launch(prefix + cfg + "_", "fetch", "FETCH_SIZE")
        write, n2, _ = per_launch(prefix + cfg + "_", "write", "WRITE_SIZE")
        if not n:
            continue
        algb = sum(4.0 * (ci + co) * alg[cfg] for ci, co in chans) / len(chans)
        out[cfg.upper()] = {
            "kernel": "the seven 3x3 launches of a step pooled (conv_wh_kernel / conv_wino*_kernel, whichever ran)", "launches_sampled": n,
            "lib_source_sha256": lib_hash,
            "source": f"profiles/{tag}_pmc_{cfg}_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KiB)",
            "fetch_size_bytes_raw": fetch * 1024, "write_size_bytes_raw": write * 1024,
            "fetch_size_bytes_x2": 2 * fetch * 1024,
            "conv_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
            "algorithmic_bytes_per_launch": algb,
            "avg_launch_ns_in_pmc_run": dur,
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 rocprofv3 reports half of wide coalesced reads; "
                    "the kernel's raw-tile DMA and weight loads are 16 B/lane); WRITE_SIZE uncalibrated, as reported; "
                    "Infinity-Cache hits are counted as traffic by these counters",
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
