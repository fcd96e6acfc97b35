# the GPU-box command list behind a profiles/<tag>_* set; raw rocprofv3 databases stay in /tmp on the box (they exceed the 64 MiB
# gpurun_out limit), only the summaries travel back.  usage: SAST_PROFILE_TAG=r02_f bash tools/refresh_profiles.sh
set -x
export TMPDIR=/tmp
TAG=${SAST_PROFILE_TAG:-r02}
O=gpurun_out/$TAG
R=$PWD
mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-batch-scan > $R/$O/kt.log 2>&1)
# (round 6: the step count of the trace = its input kernels; kernel ms and dispatches per step go into the bench line through profiles/kernel_trace_latest.json)
python tools/rocpd_stats.py /tmp/kt/kt_results.db --top 400 --out $O/kernel_trace_stats_bench_default.txt --json $O/kernel_trace_latest.json > /dev/null
cp $O/kernel_trace_latest.json profiles/kernel_trace_latest.json
timeout 120 python tools/copy_calibration.py > $O/copy_calibration.json 2>/dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o f -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/$O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o w -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/$O/pmc_write.log 2>&1)
python tools/rocpd_pmc.py --fetch /tmp/pf/f_results.db --write /tmp/pw/w_results.db --out $O/pmc_hbm_traffic.json --top 5
# the bench line of the set reads the counter summary of THIS binary (stamped with the csrc hash): install it first, then run the line
cp $O/pmc_hbm_traffic.json profiles/pmc_hbm_traffic_latest.json
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err
python tools/family_table.py $O/kernel_trace_stats_bench_default.txt $O/pmc_hbm_traffic.json $(python -c "import json;print(json.load(open('$O/kernel_trace_latest.json'))['steps'])") 8 > $O/kernel_families.md
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/sq -o sq -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/$O/sq.log 2>&1)
python tools/rocpd_sq.py /tmp/sq/sq_results.db --top 40 --out $O/sq_counters_eager_step.txt > /dev/null
# what bounds the product kernels: static resources x trace x counters x the copy calibration of THIS call (round-5 verdict item 3)
python tools/occupancy_table.py --resources profiles/r06_kernel_resources.json --trace $O/kernel_trace_stats_bench_default.txt --pmc $O/pmc_hbm_traffic.json \
  --sq $O/sq_counters_eager_step.txt --copy-tbs $(python -c "import json;print(json.load(open('$O/copy_calibration.json'))['copy_tbs'])") \
  --steps $(python -c "import json;print(json.load(open('$O/kernel_trace_latest.json'))['steps'])") --top 40 > $O/kernel_occupancy_table.md
rm -rf /tmp/kt /tmp/pf /tmp/pw /tmp/sq
timeout 300 python bench.py --res gen1 --fwd-only --steps 200 --warmup 20 > $O/bench_gen1_fwd.json 2>/dev/null
timeout 300 python bench.py --res gen1 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_gen1_train.json 2>/dev/null
timeout 300 python bench.py --seq-len 5 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_seq5.json 2>/dev/null
timeout 300 python bench.py --loss yolox --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_yolox_loss.json 2>/dev/null
timeout 300 python bench.py --seq-len 5 --loss yolox --label-every 2 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_seq5_label_sparse.json 2>/dev/null
timeout 300 python bench.py --infer --steps 100 --warmup 10 > $O/bench_infer.json 2>/dev/null
timeout 300 python bench.py --segmented --no-cpu-baseline --no-roofline > $O/bench_segmented_1gpu.json 2>/dev/null
timeout 300 python bench.py --batch 8 --no-cpu-baseline > $O/bench_b8.json 2>/dev/null
ls -la $O
