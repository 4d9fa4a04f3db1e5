"""GPU-box helper: bench.py's shard_compute under measurement variants (environment: hardware queues per process, kernel switches), and rank 0 of the
P-rank job under rocprofv3.   usage: python tools/shard_ab.py <out dir> <workload> "<P list>" [variant ...]   variant = -[,NAME=VALUE...][,prof]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

out_dir, workload, ranks = sys.argv[1], sys.argv[2], tuple(int(x) for x in sys.argv[3].split())
os.makedirs(out_dir, exist_ok=True)
steps, warm = (12, 4) if workload == "T170L60" else (30, 10)
res = {}
for v in sys.argv[4:] or ["-"]:
    parts = v.split(",")
    env = dict(p.split("=", 1) for p in parts if "=" in p)
    prof = os.path.join(out_dir, "rocprof_" + v.replace(",", "_").replace("=", "")) if "prof" in parts else None
    r = bench.shard_compute(workload, ranks=ranks, steps=steps, warmup=warm, extra_env=env, rocprof_dir=prof)
    res[v] = r
    for P in ranks:
        x = r.get(f"P={P}", {})
        if "error" in x:
            print(v, P, "ERROR", x["error"]); continue
        print(v, f"P={P}", "main", x["main_stream_ms"], "segments", x["segments_ms"], "side", x["side_stream_ms"],
              {k: round(1e3 * t, 1) for k, t in x["kernel_ms"].items()}, {k: round(1e3 * t, 1) for k, t in x["segment_ms"].items()}, flush=True)
json.dump(res, open(os.path.join(out_dir, f"shard_ab_{workload}.json"), "w"), indent=1)
