import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception:
        print(l[:300].rstrip()); continue
    if d.get("host"):
        print(d["program"], "NATIVE", {k: ("%.3g" % v if isinstance(v, float) else v) for k, v in d.items() if "per_s" in k})
        continue
    print(d["program"], "win", d.get("window"), "garble %.3g eval %.3g evalblocks %.3g groups %d fuse %s waits %s evfuse %s garble_s %.3f first %.3f" % (d["garble_gates_per_s"], d.get("eval_gates_per_s",0), d.get("eval_blocks_gates_per_s",0), d["launch_groups"], d.get("fuse"), d.get("waiting_units"), d.get("eval_fuse"), d["garble_s"], d.get("first_pass_s",0)))
