import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "step_ms", d["ms_per_step"], "dom_us", d["roofline"]["avg_launch_us"], d["phases_ms"])
    except Exception as e:
        print(f, "ERR", e)
