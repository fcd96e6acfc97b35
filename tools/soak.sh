mkdir -p gpurun_out/r03_y
run() { echo "== $*"; timeout 600 python bench.py "$@" --no-cpu-baseline --no-roofline 2>gpurun_out/r03_y/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms', 'loss', d['config']['loss_first_step'], '->', d['config']['loss'], 'finite', d['config']['grads_finite'], 'graph', d['config']['hipgraph'])" || tail -5 gpurun_out/r03_y/err.log; }
run --steps 3000 --warmup 50
run --batch 1 --steps 100 --warmup 10
run --batch 2 --steps 100 --warmup 10
run --batch 16 --steps 50 --warmup 10
run --batch 32 --steps 20 --warmup 5
run --res gen1 --batch 1 --steps 100 --warmup 10
run --res gen1 --batch 32 --steps 50 --warmup 10
run --seq-len 10 --steps 10 --warmup 3
run --loss yolox --batch 8 --steps 50 --warmup 10
run --loss yolox --seq-len 5 --label-every 3 --batch 6 --steps 20 --warmup 5
run --amp 5 --batch 2 --steps 100 --warmup 10
run --event-dtype uint8 --batch 8 --amp 0.02 --steps 50 --warmup 10
