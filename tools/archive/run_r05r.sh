#!/bin/bash
# (recreate the round-4 tree first:  git worktree add -f _r04 b9d30ca && (cd _r04 && python -c "import __graft_entry__ as g; g.build()") )
# r05r: the round-4 final tree (git worktree of b9d30ca under _r04/, its own library built from its own sources) against this tree, SAME box, alternating
mkdir -p gpurun_out/r05r
F="--steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs"
one() {   # tag, bench path, extra flags
  python $2 $F $3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  %-28s %.1f ms/batch  %.3f img/s' % ('$3', r['ms_per_step'], r['value']))"
}
{
for i in 1 2 3; do
  one "r04 (b9d30ca)" _r04/bench.py ""
  one "r05 (this)   " bench.py ""
done
one "r04 (b9d30ca)" _r04/bench.py "--config c3"
one "r05 (this)   " bench.py "--config c3"
one "r04 (b9d30ca)" _r04/bench.py "--config c4 --dtype fp8"
one "r05 (this)   " bench.py "--config c4"
one "r04 (b9d30ca)" _r04/bench.py "--config c4"
one "r05 (this)   " bench.py "--config c4 --dtype fp8c"
one "r04 (b9d30ca)" _r04/bench.py "--dtype f32x3"
one "r05 (this)   " bench.py "--dtype f32x3"
} | tee gpurun_out/r05r/r04_vs_r05_same_box.txt
