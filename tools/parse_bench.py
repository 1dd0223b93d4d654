import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline %.1f M  value_2000 %.1f M" % (j["value"]/1e6, j["value_2000"]["value"]/1e6))
for w in j.get("secondary_workloads",[]): print(w["workload"], "| %.1f M" % (w["value"]/1e6))
for w in j["other_workloads"]: print(w["workload"], "|", w.get("api","-"), "|", w.get("policy"), "| %.1f M" % (w["value"]/1e6), "%.4f ms" % w["ms_per_step"])
print("c1", j["c1"]["hip_adapter"]["steps_per_s_mean"], j["c1"]["cpu_oracle_1_thread"]["steps_per_s_mean"])
