"""stdin: bench.py's output; prints `<label> <value>` for the JSON line, or refuses it (exit 3) when the line is marked
`incomplete` (the watchdog printed it: a rank stalled) - the A/B scripts must never average such a line in.
usage: ... | python tools/line_value.py <label> [key]"""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else "value"
key = sys.argv[2] if len(sys.argv) > 2 else "value"
lines = [ln for ln in sys.stdin.read().strip().splitlines() if ln.startswith("{")]
if not lines:
    print(label, "no JSON line")
    sys.exit(2)
d = json.loads(lines[-1])
if "incomplete" in d:
    print(label, "REFUSED: line marked incomplete (%s)" % d["incomplete"])
    sys.exit(3)
print(label, d[key])
