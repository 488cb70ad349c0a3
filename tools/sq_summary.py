#!/usr/bin/env python3
"""SQ counters of a tools/gpu_round.sh `sq` stage -> profiles/sq_latest.json + a readable table.

usage: tools/sq_summary.py gpurun_out/<tag>/sq.json profiles/sq_latest.json [profiles/<name>.txt]

Per kernel (mean per launch over the bench command's launches), with the normalisations spelled out:
  clock_GHz       GRBM_GUI_ACTIVE / 8 XCDs / launch duration (the counter adds the eight XCDs' active cycles)
  valu_busy       4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles): share of all SIMD cycles of the chip in which a VALU
                  instruction is executing (SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md) — `issue_frac` of bench.py
  lds_busy        SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles); lds_conflict = SQ_LDS_BANK_CONFLICT over the same
  wave_wait       SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of a wave's life parked at s_waitcnt / s_barrier
  wave_stall      SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share stalled at issue
  wave_active     SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  waves_per_simd  SQ_WAVE_CYCLES x 4 / (1024 x kernel cycles): mean resident waves per SIMD
"""
import json
import re
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    src, out = sys.argv[1], sys.argv[2]
    js = json.load(open(src))
    rows = {}
    for k, c in js.items():
        if not isinstance(c, dict) or "GRBM_GUI_ACTIVE" not in c or "SQ_ACTIVE_INST_VALU" not in c:
            continue
        m = lambda n: c[n]["mean"] if n in c else float("nan")
        durs = [v["mean"] for n, v in c.items() if n.startswith("DURATION_NS")]
        dur = sum(durs) / len(durs)
        cyc = m("GRBM_GUI_ACTIVE") / 8.0
        if cyc <= 0 or m("SQ_WAVE_CYCLES") <= 0:
            continue
        rows[k] = {"calls": c["GRBM_GUI_ACTIVE"]["calls"], "launch_us": round(dur / 1e3, 1), "clock_GHz": round(cyc / dur, 3),
                   "valu_busy": round(4 * m("SQ_ACTIVE_INST_VALU") / (1024 * cyc), 4),
                   "salu_busy": round(4 * m("SQ_ACTIVE_INST_SCA") / (1024 * cyc), 4),
                   "lds_busy": round(m("SQ_LDS_IDX_ACTIVE") / (256 * cyc), 4), "lds_conflict": round(m("SQ_LDS_BANK_CONFLICT") / (256 * cyc), 4),
                   "wave_wait": round(m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"), 4), "wave_stall": round(m("SQ_WAIT_INST_ANY") / m("SQ_WAVE_CYCLES"), 4),
                   "wave_active": round(m("SQ_ACTIVE_INST_ANY") / m("SQ_WAVE_CYCLES"), 4),
                   "waves_per_simd": round(4 * m("SQ_WAVE_CYCLES") / (1024 * cyc), 2),
                   "insts_valu": int(m("SQ_INSTS_VALU")), "insts_salu": int(m("SQ_INSTS_SALU")), "insts_lds": int(m("SQ_INSTS_LDS")),
                   "insts_vmem": int(m("SQ_INSTS_VMEM_RD") + m("SQ_INSTS_VMEM_WR")), "waves": int(m("SQ_WAVES"))}
    issue = {}
    for g, subs in bench.GROUP_KERNELS.items():
        t = b = 0.0
        for k, r in rows.items():
            if any(re.search(r"\b" + s_ + r"\b", k) for s_ in subs):
                w = r["calls"] * r["launch_us"]
                t += w
                b += w * r["valu_busy"]
        if t:
            issue[g] = round(b / t, 4)
    res = {"_src_sha": js.get("_src_sha"), "_nseq": js.get("_nseq"), "_kn": js.get("_kn"), "_command": js.get("_command"),
           "definition": "issue_frac = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 "
                         "(time-weighted over a group's kernels)",
           "issue_frac": issue, "kernels": rows}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    hdr = f"{'kernel':62s} {'us':>8s} {'GHz':>5s} {'valu':>6s} {'salu':>6s} {'lds':>6s} {'confl':>6s} {'wait':>6s} {'stall':>6s} {'activ':>6s} {'w/simd':>6s} {'VALU insts':>11s}"
    lines = [f"# {js.get('_command')}   sources {js.get('_src_sha')}   {js.get('_nseq')} sequences, {js.get('_kn')} KeyLines per frame",
             "# " + res["definition"], hdr]
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["calls"] * kv[1]["launch_us"]):
        kk = re.sub(r"\(.*", "", k.replace("void ", "").replace("edgehip::", ""))[:62]
        lines.append(f"{kk:62s} {r['launch_us']:8.1f} {r['clock_GHz']:5.2f} {r['valu_busy']:6.3f} {r['salu_busy']:6.3f} {r['lds_busy']:6.3f} "
                     f"{r['lds_conflict']:6.3f} {r['wave_wait']:6.3f} {r['wave_stall']:6.3f} {r['wave_active']:6.3f} {r['waves_per_simd']:6.2f} {r['insts_valu']:11d}")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
