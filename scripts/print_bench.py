import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; g=d.get("roofline_large") or {}
        print(sys.argv[1], "steps/s %.0f us/step %.2f reuse %.0f | potts %.2f us frac %.3f insitu %.2f | GFP %s us frac %s" % (d["value"], d["ms_per_step"]*1e3, d["value_reuse_grad"], r["avg_launch_us"], r["frac"], r.get("avg_launch_us_in_situ", 0.0), g.get("avg_launch_us"), g.get("frac")))
