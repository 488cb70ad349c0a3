"""Turns the two rocprofv3 passes of tools/experiments/ubench_fetch (FETCH_SIZE, WRITE_SIZE; KiB per dispatch) and the
program's EXPECT lines into profiles/fetch_calibration.json: factor = bytes a perfect memory system moves / counter.
usage: fetch_calibration.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ ubench.log> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys

d, out = sys.argv[1], sys.argv[2]
expect = {}
for line in open(d + "/ubench.log"):
    m = re.match(r"EXPECT (\S+) requested=(\d+) granule32=(\d+) granule64=(\d+) granule128=(\d+)(?: reads=(\d+))?", line)
    if m:
        expect[m.group(1)] = dict(zip(("requested", "granule32", "granule64", "granule128"), map(int, m.groups()[1:5])))
        if m.group(6):   # random small reads: every read is a 64-byte request of its own (the 2 GiB working set defeats the
            expect[m.group(1)]["moved_64B_per_read"] = int(m.group(6)) * 64   # caches), repeated granules are fetched again
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{d}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0].strip()
            if r["Counter_Name"] == c:
                cnt[k][c].append(float(r["Counter_Value"]) * 1024.0)
rows = {}
for k, e in expect.items():
    c = "WRITE_SIZE" if k == "k_store4" else "FETCH_SIZE"
    v = cnt.get(k, {}).get(c)
    if not v:
        continue
    mean = sum(v) / len(v)
    rows[k] = {"counter": c, "reported_bytes": round(mean), **e, "granule64_over_reported": round(e["granule64"] / mean, 4),
               **({"moved_over_reported": round(e["moved_64B_per_read"] / mean, 4)} if "moved_64B_per_read" in e else {}),
               "granule32_over_reported": round(e["granule32"] / mean, 4), "granule128_over_reported": round(e["granule128"] / mean, 4)}
factor = {}
if "k_stream16" in rows:
    factor["stream"] = rows["k_stream16"]["granule64_over_reported"]
if "k_stream4" in rows:
    factor["stream4"] = rows["k_stream4"]["granule64_over_reported"]
if "k_stream2" in rows:
    factor["stream2"] = rows["k_stream2"]["granule64_over_reported"]
if "k_gather16" in rows:
    factor["gather16"] = rows["k_gather16"]["moved_over_reported"]
if "k_gather2" in rows:
    factor["gather2"] = rows["k_gather2"]["moved_over_reported"]
if "k_store4" in rows:
    factor["write"] = rows["k_store4"]["granule64_over_reported"]
json.dump({"what": "bytes really moved / bytes rocprofv3 reports, working set 2 GiB >> Infinity Cache, gfx950, rocprofv3 of this image "
                   "(tools/experiments/ubench_fetch.hip).  Coalesced streams: every byte of the buffer once.  Random 16-byte / 2-byte "
                   "reads at 64-byte granules: one 64-byte request per read (FETCH_SIZE = reads x 64 B to 0.2 %: the counter is exact "
                   "for them, repeated granules are fetched again because nothing that large stays cached)", "factor": factor, "kernels": rows},
          open(out, "w"), indent=1)
print(json.dumps(factor))
