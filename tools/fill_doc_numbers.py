"""Fills the @PLACEHOLDER@ numbers of DESIGN.md / README.md / INTEGRATION.md from a bench line and a shard prediction:
    python tools/fill_doc_numbers.py <bench_line.json> <shard_predict.json> <file>..."""
import json, sys


def dom_text(d):
    if not d:
        return "— (no PMC pass)"
    return "%s %.2f, useful %.2f, lanes %.2f (%.3f ms)" % (d["bound"].upper() if d["bound"] != "valu" else "VALU issue", d["frac"], d["valu_useful_frac"], d["lane_utilisation"], d["ms_per_launch"])


def main():
    line = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    sp = json.load(open(sys.argv[2]))
    r = line["roofline"]
    dom = r.get("dominant_launch")
    oc = line["other_configs"]
    v = {"C5_MRAYS": "%d" % round(line["value"], -1), "C5_TRAV": "%d" % round(line["Mrays_per_s_traversed"], -1), "C5_MS": "%.2f" % line["ms_per_step"],
         "C5_BLK": "%.2f" % line["ms_per_step_blocking"], "C5_BLKH": "%.2f" % line["ms_per_step_blocking_host"], "C5_HOST": "%.2f" % line["ms_per_step_host_output"],
         "C5_DOM": dom_text(dom), "C5_LANES": "%.2f" % (dom["lane_utilisation"] if dom else r["lane_utilisation"]),
         "C5_DOMMS": "%.2f" % (dom["ms_per_launch"] if dom else 0), "C5_DOMFRAC": "%.2f" % (dom["valu_issue_frac"] if dom else 0), "C5_USEFUL": "%.2f" % (dom["valu_useful_frac"] if dom else 0),
         "C5_HBM": "%.2f" % (dom.get("hbm_frac", 0) if dom else 0),
         "CPU1": "%.2f" % line["cpu_baseline"]["value"], "CPUN": "%.1f" % line["cpu_baseline"]["value_all_cores"]}
    for c in ("C2", "C3", "C4", "G1"):
        o = oc[c]
        v[c + "_MRAYS"] = "%d" % round(o["Mrays_per_s"], -1 if o["Mrays_per_s"] > 2000 else 0)
        v[c + "_MS"] = ("%.3f" if o["ms_per_step"] < 1 else "%.2f") % o["ms_per_step"]
        blk = o.get("ms_per_step_blocking", o.get("ms_per_step_serialised", 0))
        v[c + "_BLK"] = ("%.3f" if blk < 1 else "%.2f") % blk
        v[c + "_DOM"] = dom_text(o.get("dominant_launch"))
    for c in ("C5", "C4"):
        cfg = sp["configs"][c]
        s8 = cfg["shards"]["8_balanced"]
        v["SP_%s_WHOLE" % c] = "%.2f / %.2f" % (cfg["t_whole_ms"], cfg["period_whole_ms"])
        v["SP_%s_SHARDS" % c] = "%.2f-%.2f" % (min(s8["period_shard_ms"]), max(s8["period_shard_ms"]))
        v["SP_%s_X" % c] = "%.1f" % s8["predicted_throughput_scaling_with_fixed"]
        v["SP_%s_BAL" % c] = "%.2f" % s8["balance"]
        v["SP_%s_2" % c] = "%.1f" % cfg["shards"]["2_balanced"]["predicted_throughput_scaling_with_fixed"]
        v["SP_%s_4" % c] = "%.1f" % cfg["shards"]["4_balanced"]["predicted_throughput_scaling_with_fixed"]
    for f in sys.argv[3:]:
        s = open(f).read()
        for k, x in v.items():
            s = s.replace("@%s@" % k, x)
        left = sorted(set(__import__("re").findall(r"@[A-Z0-9_]+@", s)))
        if left:
            print(f, "unfilled:", left)
        open(f, "w").write(s)


main()
