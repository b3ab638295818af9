"""Same-box A/B of two builds of the library: python tools/ab_lib.py libA.so libB.so [bench.py args ...]
(file names inside suffix_array_amd/).  Every leg is its own process running bench.py (device-resident timing, no CPU baseline,
no end-to-end leg) with the package pointed at that file; legs alternate A B A B so that box drift shows."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
code = ("import sys, runpy; sys.path.insert(0, %r); import suffix_array_amd as sa; sa._LIB_NAME = sys.argv[1]; "
        "sys.argv = ['bench.py', '--steps', '5', '--warmup', '1', '--no-cpu-baseline', '--no-end-to-end'] + sys.argv[2:]; "
        "runpy.run_path(%r, run_name='__main__')") % (ROOT, os.path.join(ROOT, "bench.py"))
for leg in (a, b, a, b):
    out = subprocess.run([sys.executable, "-c", code, leg] + rest, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(leg, "FAILED", out.stderr[-500:]); continue
    r = json.loads(line[-1])
    print(f"{leg:40s} {r['ms_per_step']:9.3f} ms/build  verified {r.get('verified')}", flush=True)
