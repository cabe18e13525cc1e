#!/bin/bash
# rocprofv3 records of the gather kernel (BASELINE configs[1]: fused next_state + one-hot + is_solved + hash on 1M synthetic cube3
# states), run on the GPU box from the repo root: per-kernel stats and, in separate passes, FETCH_SIZE / WRITE_SIZE for the
# fp32 and the bf16 one-hot forms.  Summaries under gpurun_out/<tag>_expand_prof/summary/ (copy them into profiles/).
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${tag}_expand_prof
mkdir -p $out/summary
cd /tmp; export TMPDIR=/tmp
for oh in f32 bf16; do
  CMD="python $R/bench.py --workload expand --onehot $oh --steps 10 --warmup 2 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${oh}_stats -o ex -- $CMD > $out/${oh}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${oh}_fetch -o ex -- $CMD > $out/${oh}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${oh}_write -o ex -- $CMD > $out/${oh}_write.log 2>&1
  f=$(find $out/${oh}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/summary/${tag}_expand_1M_${oh}_kernel_stats.csv
done
cd $R
python - "$out" "$tag" <<'PY'
import collections, csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"shape": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --workload expand --onehot {f32,bf16} --steps 10 --warmup 2` "
                "(1M states per launch); HBM bytes = (2 * FETCH + WRITE) * 1024 (gfx950: FETCH_SIZE counts half the bytes of wide reads)"}
lines = ["# " + res["shape"], "%-8s %10s %16s %16s %18s" % ("onehot", "launches", "FETCH_KB(raw)", "WRITE_KB", "HBM_bytes/launch")]
for oh in ("f32", "bf16"):
    acc = collections.defaultdict(list)
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, oh, kind), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter and "expand_fused_kernel" in r["Kernel_Name"]:
                    acc[counter].append(float(r["Counter_Value"]))
    if acc["FETCH_SIZE"] and acc["WRITE_SIZE"]:
        fe = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])
        wr = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
        hb = (2 * fe + wr) * 1024
        res[oh] = {"states": 1000000, "hbm_bytes": hb, "fetch_kb_raw": fe, "write_kb": wr, "launches": len(acc["WRITE_SIZE"])}
        lines.append("%-8s %10d %16.1f %16.1f %18.0f" % (oh, len(acc["WRITE_SIZE"]), fe, wr, hb))
json.dump(res, open("%s/summary/%s_expand_pmc_traffic.json" % (out, tag), "w"), indent=1)
open("%s/summary/%s_expand_1M_pmc_traffic.txt" % (out, tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
ls -la $out/summary
