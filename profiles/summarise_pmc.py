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
import json
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


# kernel name -> timing scope of bench.py (the scopes a launch of which runs several kernels list all of them; the first is
# the one whose dispatches are counted as launches of the scope)
SCOPES = {
    "spmm_rows": ["ddx::k_spmm_packed<true", "ddx::k_spmm_lds<true"],
    "spmm_cols": ["ddx::k_spmm_packed<false", "ddx::k_spmm_lds<false"],
    "spmm_sum": ["ddx::k_sum_panels"],
    "bitplane_rows": ["ddx::k_bp_product<2, 5, 4, true", "ddx::k_bp_product<2, 4, 3, true"],
    "bitplane_cols": ["ddx::k_bp_product<2, 5, 4, false", "ddx::k_bp_product<2, 4, 3, false"],
    "residual_pack": ["ddx::k_pack_residual"],
    "knn_emit": ["ddx::k_knn_emit_bf", "ddx::k_knn_fold"],
    "knn_bound": ["ddx::k_knn_bound_bf"],
    "knn_select": ["ddx::k_knn_select", "ddx::k_knn_rescan"],
    "knn_lists": ["ddx::k_knn_tilelists"],
    "doublet_fill": ["ddx::k_doublet_fill"],
    "mirror_build": ["ddx::k_mirror_tiles", "ddx::k_mirror_count", "ddx::k_mirror_prefix"],
    "lognorm_rows": ["ddx::k_lognorm_rows", "ddx::k_lognorm_table"],
    "lognorm_cols": ["ddx::k_lognorm_csc"],
}


def traffic_table(fetch_dir, write_dir):
    """HBM GB per launch of every scope: FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE,
    both reported in KB by rocprofv3, summed over the scope's kernels and divided by the dispatches of its first kernel."""
    def totals(d, counter):
        tot, n = collections.defaultdict(float), collections.defaultdict(set)
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] == counter:
                        k = short(row["Kernel_Name"])
                        tot[k] += float(row["Counter_Value"])
                        n[k].add(row["Dispatch_Id"])
        return tot, {k: len(v) for k, v in n.items()}

    fetch, nf = totals(fetch_dir, "FETCH_SIZE")
    write, _ = totals(write_dir, "WRITE_SIZE")
    out = {}
    for scope, prefixes in SCOPES.items():
        members = [k for k in fetch if any(k.startswith(pfx) for pfx in prefixes)]
        lead = [k for k in members if k.startswith(prefixes[0])]
        launches = sum(nf.get(k, 0) for k in lead)
        if not launches:
            continue
        kb = sum(2.0 * fetch[k] + write.get(k, 0.0) for k in members)
        out[scope] = round(kb * 1024.0 / launches / 1e9, 4)
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic-json":
        # python profiles/summarise_pmc.py --traffic-json out.json fetch=<dir> write=<dir> source="..."
        kv = dict(a.split("=", 1) for a in sys.argv[3:])
        table = traffic_table(kv["fetch"], kv["write"])
        with open(sys.argv[2], "w") as fh:
            json.dump({"source": kv.get("source", ""), "unit": "GB of HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE)", "kernels": table}, fh, indent=1)
        print(table)
        return
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
