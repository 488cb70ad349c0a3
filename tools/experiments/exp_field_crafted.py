import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import test_stage_b_gpu as T
from rebvo_amd import edgehip
from oracle import oracle
import inspect
src = inspect.getsource(T.test_build_field_segments_that_round_across_a_tile_boundary)
# re-run the body up to the field download and list differences
body = src.split('f_ref, f_gpu = orc.field(0), eh.download_field(0)')[0]
body = "\n".join(l[4:] for l in body.splitlines()[1:] if not l.strip().startswith('"""') )
ns = {"np": np, "edgehip": edgehip, "pytest": None}
import re
body = body[body.index("from oracle import oracle"):]
exec(body, ns)
orc, eh, kls = ns["orc"], ns["eh"], ns["kls"]
fr, fg = orc.field(0), eh.download_field(0)
bad = (fr[..., 1] != fg[..., 1]) | ((fr[..., 1] >= 0) & (fr[..., 0] != fg[..., 0]))
ys, xs = np.nonzero(bad)
print("differing", len(ys))
for y, x in list(zip(ys, xs))[:12]:
    r, g = fr[y, x], fg[y, x]
    print((x, y), "ref", r, "gpu", g)
    for ik in {int(r[1]), int(g[1])}:
        if ik >= 0: print("    kl", ik, kls[ik]["c_p"], kls[ik]["u_m"])
