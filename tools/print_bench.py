import json,sys
d=json.load(open(sys.argv[1])); pk=d["roofline"]["per_kernel"]
print(sys.argv[1], round(d["value"],1), "Mrays/s step", round(d["ms_per_step"],2), "closest", round(pk["k_trace_closest"]["avg_launch_ms"],3), "occl", round(pk["k_trace_occluded"]["avg_launch_ms"],3), "box/tri closest", round(pk["k_trace_closest"]["node_visits_per_ray"],2), round(pk["k_trace_closest"]["tri_tests_per_ray"],2), "other ms/step", round(pk.get("other_ms_total",0)/max(d["steps"],1),2), "shadow", round(pk["k_trace_occluded"]["node_visits_per_ray"],2), round(pk["k_trace_occluded"]["tri_tests_per_ray"],2))
