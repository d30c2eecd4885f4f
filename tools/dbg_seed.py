import os, sys, time, tempfile, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ".")
    import numpy as np
    from tracy_amd import hostlib
    ip = sys.argv[2]; th = int(sys.argv[3])
    g = hostlib.Genome(ip, 15, th)
    d = np.load(sys.argv[4], allow_pickle=True)
    cons = list(d["cons"])
    for rep in range(3):
        t0 = time.perf_counter()
        sd = g.seed(cons, 50, 50, 3, 1000, th, raw=True)
        dt = time.perf_counter() - t0
    print("mapped=%s D=%s threads %d: %.0f traces/s, %.0f per thread, anchored %d" % (os.environ.get("TRACY_AMD_INDEX_MAPPED", "0"), os.environ.get("TRACY_AMD_SEED_DISTANCE", "24"), th, len(cons) / dt, len(cons) / dt / th, int(sd["status"].sum())), flush=True)
    sys.exit(0)
sys.path.insert(0, ".")
import numpy as np
from tracy_amd import hostlib
rng = np.random.default_rng(22)
n = 50_000_000
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
seq = lut[rng.integers(0, 4, size=n, dtype=np.uint8)]
d = tempfile.mkdtemp()
gp = os.path.join(d, "g.fa")
with open(gp, "wb") as f:
    f.write(b">chrSyn\n"); f.write(seq.tobytes()); f.write(b"\n")
g = hostlib.Genome(gp, 15, 16); ip = os.path.join(d, "g.tidx"); g.save(ip); g.close()
nt, mf = 32000, 1000
starts = rng.integers(0, n - mf - 50, size=nt)
comp = np.array([3, 2, 1, 0], dtype=np.uint8)
lut_inv = np.zeros(256, np.uint8); lut_inv[lut] = np.arange(4, dtype=np.uint8)
errs = np.random.default_rng(23)
cons = []
for k in range(nt):
    c = lut_inv[seq[starts[k]:starts[k] + mf]]
    if k % 2: c = comp[c[::-1]]
    flip = errs.random(mf) < 0.01
    c = np.where(flip, (c + 1) % 4, c).astype(np.uint8)
    cons.append(lut[c].tobytes())
np.savez(os.path.join(d, "c.npz"), cons=np.array(cons, dtype=object))
for th in (16,):
    for D in ("2", "3", "4", "5", "6", "8", "10"):
        for mapped in (None,):
            e = dict(os.environ, TRACY_AMD_SEED_DISTANCE=D)
            if mapped:
                e["TRACY_AMD_INDEX_MAPPED"] = mapped
            subprocess.run([sys.executable, __file__, "child", ip, str(th), os.path.join(d, "c.npz")], env=e)
import shutil; shutil.rmtree(d)
