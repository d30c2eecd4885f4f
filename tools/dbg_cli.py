"""A/B of the CLI's block pipeline on one box: serial stages (one block) vs blocks of 2000 / 1000, threads 16 / 8 per stage"""
import os, sys, time, subprocess, shutil, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch
from legs import CliLeg
leg = CliLeg(10000, 0, 1, torch.device("cuda", 0))
for cmd, n in (("align", 10000), ("decompose", 3000)):
    man, d, prep_s, b = leg.make_files(cmd, n, 1000)
    del b
    for label, env in (("serial", {"TRACY_AMD_CLI_BLOCK": "100000000"}), ("blocks2000", {}), ("blocks1000", {"TRACY_AMD_CLI_BLOCK": "1000"}),
                       ("blocks2000_t8", {"TRACY_AMD_CLI_STAGE_THREADS": "8"}), ("serial", {"TRACY_AMD_CLI_BLOCK": "100000000"}), ("blocks2000", {})):
        e = dict(os.environ, TRACY_AMD_CLI_TIMERS="1", **env)
        t0 = time.perf_counter()
        p = subprocess.run([leg.cli, cmd, "--batch", man, "-d", "0"], capture_output=True, text=True, env=e)
        dt = time.perf_counter() - t0
        tl = [ln for ln in p.stderr.splitlines() if ln.startswith("timers:")]
        print(cmd, label, "wall %.3f" % dt, tl[-1] if tl else p.stderr[-300:], flush=True)
    shutil.rmtree(d, ignore_errors=True)
