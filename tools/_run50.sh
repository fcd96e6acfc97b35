export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
run base
SAST_LIB_PATH=ab/v1.so run "v1 thin4w+k2w"
SAST_LIB_PATH=ab/v3.so run "v3 k2w"
SAST_LIB_PATH=ab/v4.so run "v4 thin4w"
run base
SAST_LIB_PATH=ab/v1.so run "v1 thin4w+k2w"
SAST_LIB_PATH=ab/v3.so run "v3 k2w"
SAST_LIB_PATH=ab/v4.so run "v4 thin4w"
