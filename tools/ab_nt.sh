#!/bin/bash
# Round 6 (verdict item 2): non-temporal stores for the activations only the backward reads (-DSAST_NT_SAVED=1 build: ab/nt1.so)
# against the product library, alternating inside ONE gpurun call: step time at the headline, B = 8 and forward-only, then the
# HBM-side bytes per step of both (two PMC passes each).   usage: bash tools/ab_nt.sh <outdir>
out=$1; mkdir -p $out
export TMPDIR=/tmp
R=$PWD
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['value'],1))"; }
for rep in 1 2 3; do
  for v in - ab/nt1.so; do
    if [ "$v" = "-" ]; then unset SAST_LIB_PATH; n=default; else export SAST_LIB_PATH=$R/$v; n=nt_saved; fi
    timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | line ${n}_headline >> $out/ab_nt.txt
    timeout 300 python bench.py --steps 100 --warmup 20 --batch 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | line ${n}_b8 >> $out/ab_nt.txt
    timeout 300 python bench.py --steps 100 --warmup 20 --res gen1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | line ${n}_gen1 >> $out/ab_nt.txt
  done
done
for v in - ab/nt1.so; do
  if [ "$v" = "-" ]; then unset SAST_LIB_PATH; n=default; else export SAST_LIB_PATH=$R/$v; n=nt_saved; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$n -o f -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/$out/pmc_fetch_$n.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$n -o w -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/$out/pmc_write_$n.log 2>&1)
  python tools/rocpd_pmc.py --fetch /tmp/pf_$n/f_results.db --write /tmp/pw_$n/w_results.db --out $out/pmc_hbm_traffic_$n.json --top 5 > /dev/null 2>&1
  rm -rf /tmp/pf_$n /tmp/pw_$n
done
unset SAST_LIB_PATH
cat $out/ab_nt.txt
python - <<PY
import json
for n in ("default", "nt_saved"):
    try:
        d = json.load(open("$out/pmc_hbm_traffic_%s.json" % n))
        print(n, {k: d[k] for k in d if "step" in k or "total" in k})
    except Exception as e:
        print(n, "pmc summary unreadable", e)
PY
