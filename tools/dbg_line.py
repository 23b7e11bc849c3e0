import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"],d["pipeline1_value"],[(o["config"],o["value"],o["pipeline1_value"]) for o in d["other_configs"]])
