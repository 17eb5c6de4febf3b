#!/usr/bin/env python3
"""Per-launch hardware counters of the step's kernels, from rocprofv3 --pmc runs, as the JSON that
bench.py reads back (profiles/rNN_traffic_<workload>.json).

    python3 tools/make_traffic_json.py WORKLOAD FETCH_DIR WRITE_DIR SQ_DIR [LDS_DIR] > profiles/r02_traffic_c3.json

FETCH_DIR / WRITE_DIR: runs with --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes: the two do not fit
one).  HBM bytes = 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes)
+ WRITE_SIZE, both reported in KiB.  SQ_DIR: a run with SQ_INSTS_VALU (wave-instructions).  LDS_DIR (may
be SQ_DIR): SQ_INSTS_LDS, SQ_ACTIVE_INST_LDS (quad-cycles), SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE.
Every value is the mean over the dispatches of the kernel in that run."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import short  # noqa: E402

STAGE_OF = {"blend_bwd_kernel": "blend_bwd", "blend_bwd_pair_kernel": "blend_bwd", "blend_fwd_kernel": "blend_fwd", "preprocess_fwd_kernel": "preprocess_fwd",
            "preprocess_bwd_kernel": "preprocess_bwd", "loss_fwd_kernel": "loss_fwd", "loss_bwd_kernel": "loss_bwd"}


def means(directory):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            stage = next((v for p, v in STAGE_OF.items() if p in k), None)
            if stage:
                a = acc[stage][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: {c: s / n for c, (s, n) in v.items()} for k, v in acc.items()}


def main(workload, fetch_dir, write_dir, sq_dir, lds_dir=None):
    fetch, write, sq = means(fetch_dir), means(write_dir), means(sq_dir)
    lds = means(lds_dir) if lds_dir else sq
    out = {"workload": workload,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* passes of `python3 bench.py --workload %s`, one "
                     "pass per run; means over the dispatches of each kernel" % workload,
           "correction": "HBM bytes = 2 x FETCH_SIZE (gfx950 tallies 128-byte read requests at 64 bytes) + WRITE_SIZE; KiB -> bytes",
           "bytes_per_launch": {}, "valu": {}, "lds": {}}
    for k in sorted(set(fetch) & set(write)):
        out["bytes_per_launch"][k] = int(round(1024 * (2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"])))
    for k in sorted(sq):
        if "SQ_INSTS_VALU" in sq[k]:
            out["valu"][k] = {"wave_instructions": int(round(sq[k]["SQ_INSTS_VALU"])),
                              "active_quad_cycles": int(round(sq[k].get("SQ_ACTIVE_INST_VALU", 0)))}
    for k in sorted(lds):
        if "SQ_INSTS_LDS" in lds[k]:
            out["lds"][k] = {"wave_instructions": int(round(lds[k]["SQ_INSTS_LDS"])),
                             "active_quad_cycles": int(round(lds[k].get("SQ_ACTIVE_INST_LDS", 0))),
                             "bank_conflict_cycles": int(round(lds[k].get("SQ_LDS_BANK_CONFLICT", 0))),
                             "idx_active_cycles": int(round(lds[k].get("SQ_LDS_IDX_ACTIVE", 0)))}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:])
