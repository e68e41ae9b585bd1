"""Build libtrunet_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(CSRC, "libtrunet_hip.so")


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INC, "*.h"))
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-I" + INC, "-I" + CSRC, "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
