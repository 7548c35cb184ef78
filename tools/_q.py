import sys,json
d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["single_frame"]["ms_per_launch"], d["ms_per_step_min_max"])
