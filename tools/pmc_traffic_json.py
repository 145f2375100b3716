"""roofline.traffic for bench.py: HBM-side bytes per launch of the dominant kernel from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
separate runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; FETCH_SIZE x2 on gfx950 for wide coalesced reads; values are KB).
python tools/pmc_traffic_json.py fetch.db write.db algorithmic_bytes > profiles/rNN_pmc_gemm_traffic.json"""
import json
import sys

from pmc_summary import per_kernel

f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
alg = float(sys.argv[3]) if len(sys.argv) > 3 else None
tot, calls, per = 0.0, 0, {}
for k, (c, kb, _d) in f.items():
    if "gemm_4w16" not in k[0]:
        continue
    wk = w.get(k, [c, 0.0, 0.0])
    tot += (2 * kb + wk[1] * c / max(wk[0], 1)) * 1024
    calls += c
    per[f"{k[0][:48]} grid {k[1]}"] = {"calls": c, "fetch_x2_MB": round(2 * kb / c / 1024, 1), "write_MB": round(wk[1] / max(wk[0], 1) / 1024, 1)}
out = {"traffic_bytes_per_launch": int(tot / max(calls, 1)), "algorithmic_bytes_per_launch": int(alg) if alg else None,
       "ratio": round(tot / max(calls, 1) / alg, 3) if alg else None, "launches_counted": calls,
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over bench.py --steps 1 --warmup 1 --no-secondary (tools/prof_round.sh); FETCH_SIZE x 2 "
                 "per the gfx950 correction of MI355X_MICROARCH.md; KB = 2^10 B", "per_kernel": per}
print(json.dumps(out, indent=1))
