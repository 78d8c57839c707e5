"""HBM traffic per dispatch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), condensed by prof_summary.py pmc.

Units and correction as MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is.
usage: pmc_traffic.py fetch_summary.csv write_summary.csv out.json
"""
import csv, json, sys

def load(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter"] == counter:
            out[r["Kernel"]] = (int(r["Dispatches"]), float(r["MeanPerDispatch"]))
    return out

def main(fetch_csv, write_csv, out_json):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fb = f.get(k, (0, 0.0))[1] * 1024.0 * 2.0
        wb = w.get(k, (0, 0.0))[1] * 1024.0
        res[k] = {"dispatches": f.get(k, w.get(k))[0], "fetch_bytes_per_dispatch": fb, "write_bytes_per_dispatch": wb,
                  "hbm_bytes_per_dispatch": fb + wb}
    json.dump({"note": "FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KiB x 1024, mean per dispatch", "kernels": res},
              open(out_json, "w"), indent=1, sort_keys=True)

if __name__ == "__main__":
    main(*sys.argv[1:4])
