#!/bin/bash
# dev: instruction counters of the two symbol-ranking kernels on the recorded block (tools/dev/symrank_bench.hip): what an item costs
# in vector / scalar / LDS instructions, summed over all 512 contexts' wavefronts and divided by the block's items
#   bash tools/dev/symrank_pmc.sh  -> gpurun_out/r06_symrank_pmc.json   (separate --pmc passes, kernel trace only)
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/srp_$c -- $REPO/build/symrank_bench $REPO/build/symrank_case.bin > /tmp/srp_$c.log 2>&1
done
python - <<PY
import glob, json, sqlite3
out = {}
for c in ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES"]:
    dbs = glob.glob("/tmp/srp_%s/**/*_results.db" % c, recursive=True)
    if not dbs:
        out[c] = "no database"; continue
    db = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]
    disp = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    rows = db.execute("select s.kernel_name, sum(p.value), count(distinct d.id) from %s p join %s d on p.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name" % (pmc, disp, sym)).fetchall()
    for name, v, n in rows:
        if "symrank" in name:
            key = "r3" if "_r3" in name else "lanes"
            out.setdefault(key, {})[c] = {"sum_over_launches": v, "launches": n}
json.dump(out, open("$REPO/gpurun_out/r06_symrank_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
