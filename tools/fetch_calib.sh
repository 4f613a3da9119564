#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for the tile kernels' gather patterns (tools/micro/fetch_calib.hip): counter bytes vs known bytes.
# GPU box; output: gpurun_out/fetch_calib.json
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o tools/micro/fetch_calib
rm -rf gpurun_out/fcal
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fcal -- ./tools/micro/fetch_calib > gpurun_out/fcal_known.json 2> gpurun_out/fcal.err
python - <<'PY'
import csv, glob, json, collections
known = json.loads(open("gpurun_out/fcal_known.json").read().strip().splitlines()[-1])
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/fcal/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
out = {"known": known, "fetch_size_bytes": {k: sum(v) / len(v) for k, v in acc.items()}}
f = out["fetch_size_bytes"]
out["ratio"] = {
    "stream16: FETCH_SIZE / bytes": f.get("stream16", 0) / known["stream16_bytes"],
    "stream4: FETCH_SIZE / bytes": f.get("stream4", 0) / known["stream4_bytes"],
    "rec_gather: FETCH_SIZE / (line bytes + id bytes)": f.get("rec_gather", 0) / (known["rec_gather_bytes"] + known["rec_gather_ids_bytes"]),
    "row_gather: FETCH_SIZE / (line bytes + id bytes)": f.get("row_gather", 0) / (known["row_gather_line_bytes"] + known["row_gather_ids_bytes"]),
    "row_gather: FETCH_SIZE / (row bytes + id bytes)": f.get("row_gather", 0) / (known["row_gather_row_bytes"] + known["row_gather_ids_bytes"]),
}
json.dump(out, open("gpurun_out/fetch_calib.json", "w"), indent=1)
print(json.dumps(out["ratio"], indent=1))
PY
