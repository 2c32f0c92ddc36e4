#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately as
MI355X_MICROARCH.md section "rocprofv3 PMC slots" requires).  Units: the counters are in KiB.  gfx950 correction
(MI355X_MICROARCH.md section HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream,
so reads are reported both raw and doubled; WRITE_SIZE is exact for 16 B/lane stores."""
import csv
import collections
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg


def main(fetch_csv, write_csv, out=None):
    fe, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    rows = []
    for k in fe:
        n, f_kib, us = fe[k]
        w_kib = wr.get(k, [0, 0.0, 0.0])[1]
        rows.append((us, k, n, f_kib * 1024 / n, w_kib * 1024 / max(1, wr.get(k, [1])[0]), us / n))
    rows.sort(reverse=True)
    lines = ["# per-launch averages.  FETCH_SIZE on gfx950 reports 1/2 of the bytes of a wide (16 B/lane) coalesced stream and an uncalibrated",
             "# fraction for other access widths (MI355X_MICROARCH.md, HBM): both the RAW and the x2-corrected read bytes are listed, with the",
             "# rate each implies.  A row whose corrected rate exceeds 6.3 TB/s (what a streaming copy achieves) is flagged '!x2': the correction",
             "# does not apply to that kernel's loads (narrower than 16 B per lane, or mostly L2 / Infinity-Cache hits) and the raw figure is the",
             "# better estimate.",
             f"{'kernel':64s} {'calls':>6s} {'avg_us':>9s} {'read_raw_MB':>11s} {'read_x2_MB':>10s} {'write_MB':>9s} {'TB/s(raw)':>9s} {'TB/s(x2)':>9s} flag"]
    for us, k, n, fb, wb, avg in rows[:30]:
        short = k if len(k) <= 64 else k[:61] + "..."
        r_raw, r_x2 = (fb + wb) / avg / 1e6, (2 * fb + wb) / avg / 1e6
        lines.append(f"{short:64s} {n:6d} {avg:9.1f} {fb/1e6:11.2f} {2*fb/1e6:10.2f} {wb/1e6:9.2f} {r_raw:9.2f} {r_x2:9.2f} {'!x2' if r_x2 > 6.3 else ''}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
