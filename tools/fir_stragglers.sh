#!/bin/bash
# Counters for the polyphase shapes furthest from the HBM roof: per shape, the matrix-core kernel's time (HIP events,
# tools/tune_fir.py) and two rocprofv3 --pmc passes over the same process (matrix-pipe busy, LDS activity / bank conflicts, wave occupancy).
#   gpurun -- bash tools/fir_stragglers.sh   ->  gpurun_out/stragglers/<dtype>_<L>_<M>.{json,pmc.json}
REPO="$(pwd)"; O="$REPO/gpurun_out/stragglers"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
CASES="${STRAGGLERS:-f32:1/8 f32:1/2 f32:2/1 f32:147/160 f32:160/147 c32:160/441 c64:160/147 c64:147/160 f32:441/160}"
for c in $CASES; do
  dt=${c%%:*}; r=${c##*:}; tag=${dt}_${r/\//_}
  echo "== $dt $r"
  export TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=3 TUNE_FIR="1,0,0"
  python $REPO/tools/tune_fir.py 2>&1 | grep "mm="; cp $REPO/gpurun_out/tune_fir.json $O/$tag.json
  rm -rf $O/p1 $O/p2
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE \
      -d $O/p1 -o s -- python $REPO/tools/tune_fir.py > $O/$tag.p1.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES \
      -d $O/p2 -o s -- python $REPO/tools/tune_fir.py > $O/$tag.p2.log 2>&1
  for p in p1 p2; do
    db=$(find $O/$p -name '*.db' | head -1)
    [ -n "$db" ] && python $REPO/tools/prof_summary.py --pmc "$db" > $O/$tag.$p.json 2>/dev/null
  done
  rm -rf $O/p1 $O/p2
  python - "$O/$tag" <<'PY'
import json, sys
b = sys.argv[1]
out = {}
for p in ("p1", "p2"):
    try:
        d = json.load(open(f"{b}.{p}.json"))
    except Exception as e:
        print("  no counters", p, e); continue
    for k, e in d.items():
        if "polyphase" in k:
            o = out.setdefault(k, {"avg_ns": e["avg_ns"], "vgpr": e["vgpr"], "lds": e["lds"], "grid": e["grid"], "wg": e["wg"], "counters": {}})
            o["counters"].update(e["counters"])
json.dump(out, open(f"{b}.pmc.json", "w"), indent=1)
for k, o in out.items():
    c = o["counters"]
    g = lambda n: c.get(n, float("nan"))
    print("  ", k[:70], "avg_us", round(o["avg_ns"] / 1e3, 1), "lds", o["lds"], "wg", o["wg"])
    print("     mfma_busy/busy_cu %.3f  mfma_insts %.3g  wave_cycles %.3g  wait_any/wave %.3f  valu_active/wave %.3f" % (
        g("SQ_VALU_MFMA_BUSY_CYCLES") / max(1, g("SQ_BUSY_CU_CYCLES")), g("SQ_INSTS_MFMA"), g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / max(1, g("SQ_WAVE_CYCLES")),
        g("SQ_ACTIVE_INST_VALU") / max(1, g("SQ_WAVE_CYCLES"))))
    print("     lds_conflict/lds_active %.3f  lds_active/busy %.3f  wait_lds/wave %.3f  lds_insts %.3g  vmem_rd %.3g" % (
        g("SQ_LDS_BANK_CONFLICT") / max(1, g("SQ_LDS_IDX_ACTIVE")), g("SQ_LDS_IDX_ACTIVE") / max(1, g("SQ_BUSY_CYCLES")), g("SQ_WAIT_INST_LDS") / max(1, g("SQ_WAVE_CYCLES")),
        g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD")))
PY
  rm -f $O/$tag.p1.json $O/$tag.p2.json
done
