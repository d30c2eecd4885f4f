"""Seeding alone (tracy_amd/host/seed.hpp getReferenceSlice over a batch) on a synthetic 50 Mb genome: traces per second by thread count
and prefetch distance (TRACY_AMD_SEED_DISTANCE), optionally under a CPU mask (`taskset -c ... python tools/exp_seed_threads.py`).
Host only; the figures of DESIGN.md section 5 come from here."""
import os
import subprocess
import sys
import tempfile
import time

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from tracy_amd import hostlib
    ip, th = sys.argv[2], int(sys.argv[3])
    g = hostlib.Genome(ip, 15, th)
    cons = list(np.load(sys.argv[4], allow_pickle=True)["cons"])
    best = 0.0
    packed, sd = hostlib.Genome.pack_consensus(cons), None
    for rep in range(5):
        t0 = time.perf_counter()
        sd = g.seed_packed(packed, 50, 50, 3, 1000, th, out=sd)
        best = max(best, len(cons) / (time.perf_counter() - t0))
    print("%s D=%s threads %d: %.0f traces/s, %.0f per thread, anchored %d" % (os.environ.get("EXP_TAG", ""), os.environ.get("TRACY_AMD_SEED_DISTANCE", "default"), th, best, best / th, int(sd["status"].sum())), flush=True)
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tracy_amd import hostlib  # noqa: E402

rng = np.random.default_rng(22)
n = 50_000_000
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
seq = lut[rng.integers(0, 4, size=n, dtype=np.uint8)]
d = tempfile.mkdtemp()
gp = os.path.join(d, "g.fa")
with open(gp, "wb") as f:
    f.write(b">chrSyn\n")
    f.write(seq.tobytes())
    f.write(b"\n")
ips = {}
for bb in [x for x in os.environ.get("EXP_BUCKET_BITS", "").split(",") if x] or [""]:
    if bb:
        os.environ["TRACY_AMD_SEED_BUCKET_BITS"] = bb
    g = hostlib.Genome(gp, 15, 16)
    ips[bb] = os.path.join(d, "g%s.tidx" % bb)
    g.save(ips[bb])
    g.close()
os.environ.pop("TRACY_AMD_SEED_BUCKET_BITS", None)
ip = ips[sorted(ips)[0]]
nt, mf = 48000, 1000
starts = rng.integers(0, n - mf - 50, size=nt)
comp = np.array([3, 2, 1, 0], dtype=np.uint8)
lut_inv = np.zeros(256, np.uint8)
lut_inv[lut] = np.arange(4, dtype=np.uint8)
errs = np.random.default_rng(23)
cons = []
for k in range(nt):
    c = lut_inv[seq[starts[k]:starts[k] + mf]]
    if k % 2:
        c = comp[c[::-1]]
    flip = errs.random(mf) < 0.01
    c = np.where(flip, (c + 1) % 4, c).astype(np.uint8)
    cons.append(lut[c].tobytes())
np.savez(os.path.join(d, "c.npz"), cons=np.array(cons, dtype=object))
masks = [m for m in os.environ.get("EXP_MASKS", "").split(";") if m]
# EXP_ENVS: configurations of the library's development knobs to compare on ONE index, e.g. "TRACY_AMD_SEED_HINT=1,TRACY_AMD_SEED_DISTANCE=12;..."
envs = [dict(kv.split("=", 1) for kv in cfg.split(",") if kv) for cfg in os.environ.get("EXP_ENVS", "").split(";") if cfg] or [{}]
for th in [int(x) for x in os.environ.get("EXP_THREADS", "16").split(",")]:
    for mask in [None] + masks:
        for extra in [dict(e, BB=bb) for bb in ips for e in envs]:
            bb = extra.pop("BB")
            cmd = [sys.executable, __file__, "child", ips[bb], str(th), os.path.join(d, "c.npz")]
            env = dict(os.environ, EXP_TAG="mask=%s bucket_bits=%s %s" % (mask or "none", bb or "24", " ".join("%s=%s" % (k.replace("TRACY_AMD_SEED_", ""), v) for k, v in extra.items())), **extra)
            if mask:
                cmd = ["taskset", "-c", mask] + cmd
            subprocess.run(cmd, env=env)
import shutil  # noqa: E402
shutil.rmtree(d)
