# runs "$@" with rocm-smi clock / power samples once a second next to it
"$@" > ${OUT:-/dev/stdout} &
PID=$!
while kill -0 $PID 2>/dev/null; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | sed -E 's/.*\(([0-9]+Mhz)\).*/\1/; s/.*\(W\): *([0-9.]+).*/\1 W/' | tr '\n' ' '; echo; sleep 1; done
