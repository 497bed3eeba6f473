import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(sys.argv[1], d["value"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["query_latency_ms"]["median"], d["mfma_queries"], d["mfma_fallback_queries"])
