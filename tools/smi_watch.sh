#!/bin/bash
# sample GPU clock / power while a command runs:  tools/smi_watch.sh out.txt -- cmd...
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|junction|Temperature" | tr '\n' ' ' ; echo; sleep 0.2; done ) > $out &
w=$!
"$@"
rc=$?
kill $w
exit $rc
