set -x
O=gpurun_out/r05_a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest.log
cp gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 600 python tools/two_graph_concurrency.py > $O/two_graph.txt 2>&1
timeout 300 python tools/two_graph_concurrency.py --fwd-only > $O/two_graph_fwd.txt 2>&1
timeout 400 python bench.py --steps 100 --warmup 20 > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.log; cat $O/two_graph.txt | tail -30; tail -c 600 $O/bench.json
