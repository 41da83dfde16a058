#!/bin/bash
# (recreate the round-3 tree first:  git worktree add -f _r03 a43a2ba && (cd _r03 && python -c "import __graft_entry__ as g; g.build()") )
# r04p: the round-3 final tree (git worktree of a43a2ba under _r03/, its own library built from its own sources) against this tree, SAME box, alternating
mkdir -p gpurun_out/r04p
F="--steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity"
one() {   # tag, bench path, extra flags
  python $2 $F $3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  %-28s %.1f ms/batch  %.3f img/s' % ('$3', r['ms_per_step'], r['value']))"
}
{
for i in 1 2 3; do
  one "r03 (a43a2ba)" _r03/bench.py ""
  one "r04 (this)   " bench.py "--no-other-configs"
done
one "r03 (a43a2ba)" _r03/bench.py "--config c3"
one "r04 (this)   " bench.py "--config c3 --no-other-configs"
one "r03 (a43a2ba)" _r03/bench.py "--config c4"
one "r04 (this)   " bench.py "--config c4 --dtype fp8 --no-other-configs"
one "r04 (this)   " bench.py "--config c4 --no-other-configs"
one "r03 (a43a2ba)" _r03/bench.py "--dtype f32x3"
one "r04 (this)   " bench.py "--dtype f32x3 --no-other-configs"
} | tee gpurun_out/r04p/r03_vs_r04_same_box.txt
