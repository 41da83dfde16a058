mkdir -p gpurun_out/r03j
python -m pytest tests -q -m gpu 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py > gpurun_out/r03j/bench_default.json 2> gpurun_out/r03j/bench_default.err; head -c 700 gpurun_out/r03j/bench_default.json; echo
