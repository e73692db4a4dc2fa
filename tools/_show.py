import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["per_rank_seconds"], d["config"]["parallelism"], d.get("single_stream_images_per_sec"), d.get("e2e_in_flight_images_per_sec"))
