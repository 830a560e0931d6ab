import json,sys
for f in sys.argv[1:]:
    for line in open(f):
        if line.startswith("{"):
            d=json.loads(line)
            print(f, d["value"], d["ms_per_step"])
            for k,v in d.get("hbm_kernels",{}).items():
                print("   ",k, v.get("avg_launch_ms"), v.get("frac_of_8TBps"))
