"""HBM bytes per launch of the core kernels from rocprofv3 --pmc passes (rocpd sqlite) -> the JSON `bench.py` reads.

    python3 scripts/pmc_traffic_json.py <dir with fetch/ and write/ pass outputs> <core|stress> <out.json> [extra dirs...]

FETCH_SIZE and WRITE_SIZE come from SEPARATE passes (they do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots);
both are reported in KiB.  gfx950 correction (same guide, HBM section): FETCH_SIZE reports half the bytes of a wide coalesced
streaming read, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; kernels listed in AS_REPORTED read 4 B per lane in short
row segments (an uncalibrated width) and keep FETCH_SIZE as reported."""
import collections, glob, json, sqlite3, sys

STAGES = {
    "core": {"wpmi": ["wpmi_slice_kernel"], "topk": ["neuron_topk_fast_kernel"], "gemm": ["gemm_nt_f32_kernel", "gemm_nt_f32_dma_kernel", "normalize_rows_kernel"],
             "softmax": ["row_softmax_lds_kernel"], "logsumexp": ["lse_panel_kernel"], "row_topk": ["row_topk_kernel", "row_topk_short_kernel"]},
    "stress": {"wpmi": ["wpmi_bf16_kernel"], "topk": ["neuron_topk_fast_kernel"],   # incl. K6's long rows (same kernel template)
               "gemm": ["gemm_nt_bf16_exp", "normalize_to_bf16_kernel", "rowsum_finish_kernel"],
               "logsumexp": ["lse_panel_kernel"], "row_topk": ["row_topk_kernel", "neuron_topk_fast_kernel<256"]},
}
AS_REPORTED = {"core": ("lse_panel_kernel",),   # C = 763: an odd pitch, 4 bytes per lane
               "stress": ()}                   # C = 10 000: the panel kernel moves 16 bytes per lane like everything else


def per_kernel(root, counter):
    """{kernel name: (mean counter value per dispatch summed over XCD instances, dispatches, mean duration us)}"""
    out = {}
    for path in sorted(glob.glob(root + "/**/*.db", recursive=True)):
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        ki, ci, vi, di = (cols.index(c) for c in ("kernel_name", "counter_name", "value", "dispatch_id"))
        st, en = cols.index("start"), cols.index("end")
        acc, dur = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(dict)
        for r in db.execute("select * from counters_collection"):
            if r[ci] != counter:
                continue
            acc[r[ki]][r[di]] += r[vi]
            dur[r[ki]][r[di]] = r[en] - r[st]
        for k, d in acc.items():
            out[k] = (sum(d.values()) / len(d), len(d), sum(dur[k].values()) / len(d) / 1e3)
    return out


def main():
    root, which, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, write = per_kernel(root + "/fetch", "FETCH_SIZE"), per_kernel(root + "/write", "WRITE_SIZE")
    res = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) -- python3 bench.py --config %s "
                   "--steps 3 --warmup 1 --no-cpu-baseline (scripts/r05_pmc.sh); mean per dispatch in KiB as reported; hbm_bytes = "
                   "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads); "
                   "%s  A stage = the sum over its kernels, each "
                   "times its launches per pass of the core." % (which, "lse_panel_kernel's 4-byte-per-lane loads (odd pitch) keep FETCH_SIZE as reported."
                                                                 if which == "core" else "")}
    for stage, pats in STAGES[which].items():
        tot_f = tot_w = hbm = 0.0
        names = []
        for k in fetch:
            if not any(p in k for p in pats):
                continue
            if stage == "topk" and "neuron_topk_fast_kernel<256" in k and which == "stress":
                continue       # the 256-thread class at this shape is K6's long-row path
            f, n, us = fetch[k]
            w = write.get(k, (0.0, 0, 0.0))[0]
            n_runs = max(1, min(v[1] for kk, v in fetch.items() if "wpmi" in kk))   # passes of the core = K4 launches
            per_run = n / n_runs
            mul = 1.0 if any(a in k for a in AS_REPORTED[which]) else 2.0
            tot_f += f * per_run
            tot_w += w * per_run
            hbm += (mul * f + w) * 1024 * per_run
            short = k.split("(anonymous namespace)::", 1)[-1].split("(")[0]   # (argument types may name the namespace again)
            names.append("%s x%g (%.1f us)" % (short, per_run, us))
        if names:
            res[stage] = {"kernel": " + ".join(names), "FETCH_SIZE_KiB": tot_f, "WRITE_SIZE_KiB": tot_w, "hbm_bytes": int(hbm)}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
