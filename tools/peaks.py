#!/usr/bin/env python
"""Re-derive the roofline denominators on the box: builds tools/peaks.hip with hipcc and runs it.

  python tools/peaks.py [out.txt]        prints the probe's table; the last line is a JSON object {probe: {value, unit}}

bench.py reads the committed result (profiles/rNN/peaks.json) for `roofline.achievable`."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "peaks_probe")
SRC = os.path.join(HERE, "peaks.hip")


def main():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-o", EXE, SRC])
    out = subprocess.run([EXE], stdout=subprocess.PIPE, check=True).stdout.decode()
    sys.stdout.write(out)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(out)
        js = json.loads(out.strip().splitlines()[-1])
        with open(os.path.splitext(sys.argv[1])[0] + ".json", "w") as f:
            json.dump(js, f, indent=1)


if __name__ == "__main__":
    main()
