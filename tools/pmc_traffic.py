"""HBM traffic per dispatch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), condensed by prof_summary.py pmc.

Units and correction as MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is.
usage: pmc_traffic.py fetch_summary.csv write_summary.csv out.json
"""
import csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hashes():
    """sha256 of the kernel sources the bench line's `roofline.traffic` speaks about: bench.py quotes this file only while they are unchanged"""
    out = {}
    for name in ("gru.hip", "common.hpp"):
        with open(os.path.join(ROOT, "chainer-speech-recognition_amd", "csrc", name), "rb") as f:
            out[name] = hashlib.sha256(f.read()).hexdigest()
    return out


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
    json.dump({"note": "FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KiB x 1024, mean per dispatch", "kernels": res,
               "source_sha256": source_hashes()},
              open(out_json, "w"), indent=1, sort_keys=True)

if __name__ == "__main__":
    main(*sys.argv[1:4])
