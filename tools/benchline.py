"""Print the figures of a bench.py JSON line that are compared from run to run (one short line each)."""
import json
import sys

line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = line.get("roofline", {})
print("value %.3f G/s  us_per_batch %.2f" % (line["value"] / 1e9, line["config"].get("us_per_batch", 0)))
for k in ("fused_instantiation", "plain_scorer", "configs1_launch"):
    v = r.get(k)
    if v:
        print("%-20s kernel_ms %.4f frac %.3f" % (k, v.get("kernel_ms", 0), v.get("frac", 0)))
for leg in ("tucker", "pgpr", "cafe", "sharded"):
    v = line.get(leg)
    if not v:
        continue
    if "error" in v:
        print("%s ERROR %s" % (leg, v["error"]))
    elif leg == "tucker":
        print("tucker fwd %.3f ms  fwd+bwd %.3f ms" % (v["forward"]["ms"], v["forward_backward"]["ms"]))
    else:
        print("%s ms_per_step %.3f value %.4g" % (leg, v["ms_per_step"], v["value"]))
