import numpy as np, sys, subprocess, os, json, tempfile
sys.path.insert(0, "/root/repo")
from rebvo_amd import edgehip, synth
from tests.helpers import write_global_config
W,H=376,240
def tri(k, n):
    p = 2 * (n - 1); k %= p
    return k if k < n else p - k
tmp=tempfile.mkdtemp()
n_obj, n_fr, pool = 2, 7, 6
frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=13)]
np.stack(frames).tofile(tmp+"/frames.rgb24")
write_global_config(tmp+"/cfg", edgehip.euroc_params(W, H))
r = subprocess.run(["/root/repo/rebvo_amd/lib/surface_replay", tmp+"/cfg", tmp+"/frames.rgb24", str(pool), str(n_obj), str(n_fr), "1.0", "0.05", "--group", "snap", "--snapshot-at", "4"], capture_output=True, text=True, cwd=tmp)
print(r.stdout[-600:])
snap=open(tmp+"/Snap0.ppm","rb").read()
head = f"P6\n{W} {H} 255\n".encode()
img = np.frombuffer(snap[len(head):], np.uint8).reshape(H, W, 3)
for j,f in enumerate(frames):
    print(j, int((img!=f).sum()))
print("rows equal to frame4:", [(int((img[y]!=frames[4][y]).sum())) for y in range(0,H,40)])
