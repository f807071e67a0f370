"""Regenerates profiles/INDEX.md: file -> what it shows -> superseded by (same measurement, later state).  python profiles/tools/make_index.py"""
import os
import re

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = [
    (r"_kernel_stats_1stream\.csv$", "rocprofv3 --kernel-trace --stats of `DDX_STREAMS=1 bench.py` (every kernel alone on the GPU: the per-kernel averages quoted in DESIGN and `bench.py`'s roofline)"),
    (r"_kernel_stats.*\.csv$", "rocprofv3 --kernel-trace --stats of the default (multi-context) bench run"),
    (r"_pmc_counters\.txt$", "FETCH_SIZE / WRITE_SIZE / TCC hit-miss per kernel from separate --pmc passes (source of `roofline.traffic`)"),
    (r"_pmc_sq.*\.txt$", "SQ counters (VALU / LDS / MFMA busy, bank conflicts) of the named kernels"),
    (r"_launch_sequence\.txt$", "kernel launches of the last iterations of a fit in start order (kernel trace)"),
    (r"_bench_line.*\.json$", "one `bench.py` JSON line measured by the builder on a gpurun box"),
    (r"_configs\.txt$", "`bench.py` at the other BASELINE shapes (configs[1], leiden, louvain + scaling, configs[3] on one GPU)"),
    (r"_busy_timeline\.txt$", "union of the kernel intervals of all contexts over one fit in 5 ms windows"),
    (r"_gaps\.txt$", "idle gaps between the kernels of a fit"),
    (r"_concurrency\.txt$", "kernels in flight / share of the window without a GPU-filling kernel"),
    (r"_cpu_full\.json$", "CPU oracle at the full headline size (two iterations, scaled to n_iters)"),
]
SPECIAL = {
    "INDEX.md": "this table",
    "pmc_traffic.json": "per-kernel HBM bytes summarised from the latest *_pmc_counters.txt by summarise_pmc.py; read by bench.py",
    "summarise_pmc.py": "tool: *_pmc_counters.txt -> pmc_traffic.json",
    "collect.sh": "tool: regenerates kernel stats, launch sequence and PMC passes for a tag",
    "pmc_sq.sh": "tool: SQ counter passes",
    "timeline.sh": "tool: busy timeline",
    "HISTORY.md": "analyses behind decisions of rounds 1-5 that DESIGN.md no longer carries (rejected product kernels, kNN history, priced-and-not-built items)",
    "r06_mx_notes.txt": "round 6: the bit-plane products on the MX matrix instruction (FP4 x FP6): probe, three kernel generations, stage sweep, ablations; the digit schedule bp_digits_early and why it stays off",
    "r06e_fp6_probe.txt": "round 6: layout check and MAC rates of v_mfma_f32_32x32x64_f8f6f4 against int8 under random digits (profiles/tools/mfma_fp6_probe.hip)",
    "r06i_probe_digit_patterns.txt": "round 6: the same probe with non-negative 7-bit / 4-bit digits (power-bound int8 kernel: +5 - 12 %)",
    "r06e_digits_ab.txt": "round 6: fits with bp_digits_early = 0 (A) / 3 (B) / 2 (C), same box, alternating",
    "r06e_digits_test.txt": "round 6: PCA scores against the float64 run with 4 / 3 / 2 digits in the early power iterations (configs[1])",
    "r06f_mx_stage_sweep.txt": "round 6: matrix-core product kernels' launch time against the stages per chunk, int8 and MX form (fixed cost per launch vs per stage)",
    "r06f_mx_ablation.txt": "round 6: compile-time ablations of the MX product kernel v2 (copies / bitmap copies / matrix instructions taken out)",
    "r06g_mx_ablation.txt": "round 6: the same for v3 (expansion interleaved with the matrix instructions)",
    "r06g_mx_ab.txt": "round 6: fits with bp_format=int8 (A) / mx6 (B), same box, alternating",
    "r06_share_times.txt": "round 6: the busiest rank's share of 10 iterations at 2 / 4 / 8 GPUs run on one GPU (DESIGN section 6's projection)",
    "r06h_gpu_tests_summary.txt": "round 6: last lines of the full `pytest -m gpu` run at HEAD (143 passed)",
    "r05_bitplane_notes.txt": "round 5: bit-plane product kernel (matrix pipe busy 82 %, power-bound), packed residue ablations (section 7 of the file)",
    "r05_block_lanczos.txt": "round 5: pseudocount = 1 block Lanczos schedule, 70 ms per PCA at configs[1]",
    "r04_bitplane_notes.txt": "round 4: first bit-plane version, not faster",
    "r04_knn_notes.txt": "round 4: kNN cells / tile lists / work-item emit measurements",
    "r04_knn_prune_study.txt": "round 4: offline pruning study (boxes vs centre directions)",
    "r04_block_lanczos.txt": "round 4: first block Lanczos",
    "r04_spmm_dpp_proto.txt": "round 4: DPP register-broadcast prototype of the sparse product (rejected)",
    "r06_host_wait_hang.txt": "round 6: stack of the process that hung at exit with hipDeviceScheduleBlockingSync, and the fix (library-side sleeping polls)",
    "r06_host_wait_cpu.txt": "round 6: CPU time per headline fit with host_wait = spin | block (cgroup cpu.stat)",
    "r06_scaled_route.txt": "round 6: standard_scaling on the bit-plane route -- accuracy against the float64 oracle, demoted columns, timings",
}
SUPERSEDES = {"r04_bitplane_notes.txt": "r05_bitplane_notes.txt", "r04_block_lanczos.txt": "r05_block_lanczos.txt"}


def rnd(f):
    m = re.match(r"r0(\d)", f)
    return int(m.group(1)) if m else 0


files = sorted(f for f in os.listdir(HERE) if os.path.isfile(os.path.join(HERE, f)))
rows = []
for f in files:
    d = SPECIAL.get(f)
    if not d:
        for pat, desc in KINDS:
            if re.search(pat, f):
                d = desc
                break
    if not d:
        d = f"notes / measurement of round {rnd(f)} (quoted in DESIGN.md or profiles/HISTORY.md)" if rnd(f) else "see the file's header"
    rows.append((f, d))
latest = {}
for f, _ in rows:
    m = re.match(r"(r0\d[a-z]?)_(.*)", f)
    if m:
        latest[m.group(2)] = max(latest.get(m.group(2), ""), m.group(1))
out = ["# profiles/ index", "",
       "File -> what it shows -> superseded by (the same measurement at a later state).  Prefixes `r01` ... `r06` are rounds; a letter orders the states inside a round.  "
       "Regenerate with `python profiles/tools/make_index.py`; the scripts under `profiles/tools/` are indexed in `profiles/tools/README.md`.", "",
       "| file | what it shows | superseded by |", "|---|---|---|"]
for f, d in rows:
    m = re.match(r"(r0\d[a-z]?)_(.*)", f)
    sup = SUPERSEDES.get(f, "")
    if m and latest[m.group(2)] != m.group(1):
        sup = f"{latest[m.group(2)]}_{m.group(2)}"
    out.append(f"| `{f}` | {d} | {('`' + sup + '`') if sup else ''} |")
open(os.path.join(HERE, "INDEX.md"), "w").write("\n".join(out) + "\n")
print(len(rows), "files indexed")
