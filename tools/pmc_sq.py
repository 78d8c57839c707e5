"""Condense a rocprofv3 --pmc SQ pass over bench.py into profiles/<tag>_pmc_sq.json / .csv: per kernel, mean per dispatch of every
counter and the derived fractions north_star asks for ("MFMA utilisation for the GEMM blocks against gfx950 peak").

    python3 tools/pmc_sq.py <counter_collection.csv> <out.json> <out.csv>

Units (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over the chip; SQ_WAVE_CYCLES /
SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the
dispatch was in flight.  mfma_busy_frac = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the fraction of the
chip's matrix-pipe cycles the kernel kept busy while it ran (1.0 = every SIMD issuing MFMAs back to back)."""
import collections
import csv
import json
import sys

CUS, SIMDS, XCDS = 256, 4, 8
KEEP = ("gemm_nt_8ph_kernel", "gemm_nt_8pp_kernel", "gemm_nt_8pn_kernel", "gemm_tn_8ph_kernel", "specgram512_kernel", "deltas_tile_kernel", "merge_dirs", "gemm_nt256p_kernel", "gemm_nt256_kernel", "gemm_nt_kernel", "gemm_nt_wide_kernel", "gemm_tn_kernel", "gemm_tn_vec_kernel", "gemm_tn256_kernel",
        "lattice_kernel", "fwd_persistent_io_kernel", "bwd_ps_kernel", "bwd_wide_kernel", "ctcln::bwd_kernel", "rows_kernel",
        "fwd_rows_f32_kernel", "sru::", "maxout2_pool", "adam_ctl_kernel", "convf::", "conv_direct")


def main(path, out_json, out_csv):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("void ", "")
        name = name.split("(")[0]
        c = agg[name][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"])
        c[1] += 1
    kernels = {}
    for name, counters in sorted(agg.items()):
        if not any(k in name for k in KEEP):
            continue
        mean = {cn: sm / n for cn, (sm, n) in counters.items()}
        n = max(v[1] for v in counters.values())
        e = {"dispatches": n, "mean_per_dispatch": {k: round(v, 1) for k, v in sorted(mean.items())}}
        gui = mean.get("GRBM_GUI_ACTIVE")
        if gui:
            chip_simd_cycles = gui / XCDS * CUS * SIMDS
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
                e["mfma_busy_frac"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / chip_simd_cycles, 4)
            if "SQ_LDS_IDX_ACTIVE" in mean:
                e["lds_active_frac_of_cu_cycles"] = round(mean["SQ_LDS_IDX_ACTIVE"] / (gui / XCDS * CUS), 4)
        wave = mean.get("SQ_WAVE_CYCLES")
        if wave:
            for cn, key in (("SQ_WAIT_ANY", "wave_parked_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "issuing_frac")):
                if cn in mean:
                    e[key] = round(mean[cn] / wave, 4)
        if mean.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(mean.get("SQ_LDS_BANK_CONFLICT", 0.0) / mean["SQ_LDS_IDX_ACTIVE"], 4)
        kernels[name] = e
    json.dump({"source": "rocprofv3 --pmc (one SQ pass + GRBM_GUI_ACTIVE) over `python3 bench.py --steps 2 --warmup 1 --no-census "
                         "--no-cpu-baseline --no-extra`; tools/profile_sq.sh", "units": __doc__.split("Units")[1].strip(),
               "kernels": kernels}, open(out_json, "w"), indent=1)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Dispatches", "mfma_busy_frac", "wave_parked_frac", "issue_stall_frac", "issuing_frac", "lds_active_frac_of_cu_cycles",
                    "lds_bank_conflict_frac"])
        for k, e in kernels.items():
            w.writerow([k, e["dispatches"]] + [e.get(c, "") for c in ("mfma_busy_frac", "wave_parked_frac", "issue_stall_frac", "issuing_frac",
                                                                     "lds_active_frac_of_cu_cycles", "lds_bank_conflict_frac")])


if __name__ == "__main__":
    main(*sys.argv[1:4])
