"""bench.py's heterogeneous sequence 5 on the GPU box: free-running and teacher-forced replays of the device path against the
reference running on this host (oracle/teacher.py), as JSON — the evidence file behind DESIGN.md section 5."""
import json
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from helpers import hetero_sequence
from oracle import oracle, teacher
from rebvo_amd import edgehip

W, H = 752, 480
seq = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 14
out = {"host_cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(" \t:"), "sequence": seq, "frames": nf}
for forced in (False, True):
    orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=1, nslots=3)
    r = teacher.teacher_forced_replay(eh, orc, hetero_sequence(seq), nf, forced=forced)
    eh.close()
    orc.close()
    out["teacher_forced" if forced else "free_running"] = {
        "dV_per_frame": [float(f"{x:.3e}") for x in r["dV"]], "dW_per_frame": [float(f"{x:.3e}") for x in r["dW"]],
        "outside_tolerance": r["outside_tolerance"], "knife_edge_frames": r["knife_edge_frames"]}
print(json.dumps(out, indent=1))
