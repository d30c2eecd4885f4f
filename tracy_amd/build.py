"""Build the gfx950 shared library in-tree: tracy_amd/lib/libtracy_hip.so (hipcc cross-compiles
without a GPU).  The library is the product; it has no CPU path."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libtracy_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"] + os.environ.get("TRACYHIP_CXXFLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def deps():
    out = []
    for f in os.listdir(CSRC):
        out.append(os.path.join(CSRC, f))
    out.append(os.path.join(os.path.dirname(HERE), "include", "tracy_hip.h"))
    return out


def stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or stale(obj, deps()):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or stale(SO, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SO




HOST_SO = os.path.join(LIBDIR, "libtracy_host.so")


def build_host(force=False, verbose=False):
    """host-side C++ (basecall / createProfile / synthetic workloads): plain g++, no GPU code"""
    os.makedirs(LIBDIR, exist_ok=True)
    hdir = os.path.join(HERE, "host")
    srcs = [os.path.join(hdir, f) for f in os.listdir(hdir)]
    if force or stale(HOST_SO, srcs):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-Wall",
               "-o", HOST_SO, os.path.join(hdir, "tracy_host_capi.cpp"), "-lz"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_SO


CLI = os.path.join(HERE, "bin", "tracy_amd_cli")


def build_cli(force=False, verbose=False):
    """`tracy align` command line over the C ABI (tracy_amd/cli): g++ host code linked against libtracy_hip.so"""
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    hdir = os.path.join(HERE, "host")
    src = os.path.join(HERE, "cli", "tracy_amd_cli.cpp")
    srcs = [src, os.path.join(HERE, "cli", "assemble_cli.inc"), os.path.join(HERE, "cli", "consensus_cli.inc"), SO, os.path.join(os.path.dirname(HERE), "include", "tracy_hip.h")]
    srcs += [os.path.join(hdir, f) for f in os.listdir(hdir)]
    if force or stale(CLI, srcs):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-o", CLI, src, "-L" + LIBDIR, "-ltracy_hip",
               "-Wl,-rpath,$ORIGIN/../lib", "-lz", "-pthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI


MSA_SO = os.path.join(LIBDIR, "libtracy_msa.so")


def build_msa(force=False, verbose=False):
    """progressive multiple alignment over the C ABI (tracy_amd/host/msa.hpp): g++ host code linked against libtracy_hip.so"""
    hdir = os.path.join(HERE, "host")
    src = os.path.join(hdir, "msa_capi.cpp")
    srcs = [src, SO, os.path.join(hdir, "msa.hpp"), os.path.join(hdir, "tracy_host.hpp"), os.path.join(os.path.dirname(HERE), "include", "tracy_hip.h")]
    if force or stale(MSA_SO, srcs):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-o", MSA_SO, src, "-L" + LIBDIR, "-ltracy_hip",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return MSA_SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
    print(build_cli(force="--force" in sys.argv, verbose=True))
    print(build_msa(force="--force" in sys.argv, verbose=True))
