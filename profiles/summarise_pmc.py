"""Per-kernel averages of rocprofv3 counter-collection CSVs.

usage: python profiles/summarise_pmc.py <tag>=<dir> [<tag>=<dir> ...] > profiles/rNNx_pmc_counters.txt

Each <dir> is the -d directory of one `rocprofv3 --kernel-trace --pmc ...` pass (counters are collected in
separate passes, never together with the sys/runtime/hip/hsa trace domains).  FETCH_SIZE / WRITE_SIZE are
reported in KB by rocprofv3; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced
reads (MI355X_MICROARCH.md, HBM section), so the "x2" column is the number to compare with a byte model.
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    m = re.match(r"_ZN3ddx(\d+)", name)          # names rocprofv3 could not demangle (bf16 template arguments)
    if m:
        n = int(m.group(1))
        return "ddx::" + name[m.end():m.end() + n]
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def main():
    for arg in sys.argv[1:]:
        tag, d = arg.split("=", 1)
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    disp[k].add(row["Dispatch_Id"])
        for k in sorted(acc):
            n = max(len(disp[k]), 1)
            vals = {c: round(v / n, 1) for c, v in sorted(acc[k].items())}
            extra = ""
            if "FETCH_SIZE" in vals:
                extra = f" fetch_x2_MB {vals['FETCH_SIZE'] * 2 / 1024:.1f}"
            if "WRITE_SIZE" in vals:
                extra += f" write_MB {vals['WRITE_SIZE'] / 1024:.1f}"
            if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
                tot = vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]
                extra += f" l2_hit {vals['TCC_HIT_sum'] / tot:.3f}" if tot else ""
            print(f"{tag} {k} {vals} dispatches {n}{extra}")


if __name__ == "__main__":
    main()
